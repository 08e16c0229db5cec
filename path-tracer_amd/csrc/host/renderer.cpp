#include "renderer.hpp"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <thread>

#include "flatten.hpp"

// ---------------------------------------------------------------- PpmWriter
void PpmWriter::write(Pixel const& p) {  // framework/ppmwriter.cpp:33-45
	std::size_t buf_pos = width_ * (height_ - 1 - p.y) + p.x;
	std::size_t pos = 3 * buf_pos;
	if (pos + 2 > data_.size() || (int) buf_pos < 0) {
		std::cerr << "Fatal Error PpmWriter::write(Pixel p) : Critical write position\n";
	} else {
		data_[pos] = (unsigned int) std::max(0.0, std::min(255.0 * p.color.r, 255.0));
		data_[pos + 1] = (unsigned int) std::max(0.0, std::min(255.0 * p.color.g, 255.0));
		data_[pos + 2] = (unsigned int) std::max(0.0, std::min(255.0 * p.color.b, 255.0));
	}
}
void PpmWriter::set_rows(const unsigned char* rgb8) {
	for (std::size_t i = 0; i < data_.size(); ++i) data_[i] = rgb8[i];
}
void PpmWriter::save(std::string const& file) {
	file_ = file;
	save();
}
void PpmWriter::save() {  // framework/ppmwriter.cpp:55-76: the same text, put together by the library's writer
	std::vector<unsigned char> bytes(data_.size());
	for (std::size_t i = 0; i < data_.size(); ++i) bytes[i] = (unsigned char) data_[i];  // (write() stores 0..255)
	if (ptx_write_ppm(file_.c_str(), bytes.data(), (uint32_t) width_, (uint32_t) height_) != PTX_OK)
		std::cerr << "PpmWriter::save : " << ptx_last_error() << "\n";  // (the reference ignores a file it cannot open)
}

// ---------------------------------------------------------------- Renderer
Renderer::Renderer(unsigned w, unsigned h, std::string const& file, unsigned aa_steps, unsigned max_ray_bounces)
    : width_(w), height_(h), color_buffer_(w * h, Color{0.0, 0.0, 0.0}), filename_(file), ppm_(width_, height_), aa_steps_(aa_steps),
      max_ray_bounces_(max_ray_bounces) {}

Renderer::~Renderer() { drop_scene(); }

void Renderer::drop_scene() {
	for (PtxScene* s : dev_scenes_) ptx_scene_destroy(s);
	dev_scenes_.clear();
	kept_root_.reset();
	kept_print_ = 0;
}

bool Renderer::select_devices(int first, int n, int visible, unsigned long long samples, int* out_first, int* count) {
	if (visible < 1 || first < 0 || first >= visible || n < 0) return false;
	int k = n == 0 ? visible - first : n;
	if (first + k > visible) return false;
	if (samples < (unsigned long long) k) k = (int) (samples < 1 ? 1 : samples);  // every GPU renders at least one sample of every pixel
	*out_first = first;
	*count = k;
	return true;
}

namespace {
// FNV-1a over the values render() reads outside the shape tree: a camera moved or a light changed between two
// render() calls of a kept scene must not be answered from the copy on the GPUs
unsigned long long fingerprint(Scene const& scene) {
	unsigned long long h = 1469598103934665603ull;
	auto mix = [&](const void* p, size_t n) {
		const unsigned char* b = static_cast<const unsigned char*>(p);
		for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
	};
	auto mixf = [&](float v) { mix(&v, sizeof v); };
	auto mixc = [&](Color const& c) { mixf(c.r); mixf(c.g); mixf(c.b); };
	auto mixv = [&](vec3 const& v) { mixf(v.x); mixf(v.y); mixf(v.z); };
	mixf(scene.camera.fov_x); mixv(scene.camera.position); mixv(scene.camera.direction); mixv(scene.camera.up);
	mixc(scene.ambient.intensity);
	for (auto const& l : scene.lights) { mixv(l.position); mixc(l.intensity); }
	for (auto const& [name, m] : scene.materials) {
		mix(name.data(), name.size());
		if (m) { mixc(m->ka); mixc(m->kd); mixc(m->ks); mixf(m->m); mixf(m->glossy); mixf(m->opacity); mixf(m->ior); }
	}
	const unsigned long long counts[2] = {scene.lights.size(), scene.materials.size()};
	mix(counts, sizeof counts);
	return h ? h : 1;
}
}  // namespace

void Renderer::render(Scene const& scene) {
	auto start = std::chrono::steady_clock::now();
	auto raise = [&](std::string msg) {
		drop_scene();
		error_ = std::move(msg);
		throw static_cast<const char*>(error_.c_str());
	};
	setup_seconds_ = 0;
	const unsigned long long print = fingerprint(scene);
	if (!(keep_ && !dev_scenes_.empty() && !kept_root_.expired() && kept_root_.lock() == scene.root && kept_print_ == print)) {
		drop_scene();
		FlatScene flat;
		try {
			flat = flatten_scene(scene);
		} catch (SceneError const& e) {
			raise(e.message);
		}
		const int visible = ptx_device_count();
		if (visible < 1) raise("no gfx950 GPU: this Renderer has no CPU path");
		int first = 0, n = 0;
		if (!select_devices(first_device_, n_devices_, visible, (unsigned long long) aa_steps_ * aa_steps_, &first, &n)) raise("the GPUs asked for are not there");
		// one copy of the scene per GPU, created side by side (BVH build and upload of one do not wait for the other's)
		dev_scenes_.assign((size_t) n, nullptr);
		std::vector<std::string> errors((size_t) n);
		auto create = [&](int k) {
			if (ptx_scene_create(&flat.desc, first + k, 0, &dev_scenes_[(size_t) k]) != PTX_OK) errors[(size_t) k] = ptx_last_error();
		};
		{
			std::vector<std::thread> threads;
			for (int k = 1; k < n; ++k) threads.emplace_back(create, k);
			create(0);
			for (auto& t : threads) t.join();
		}
		for (int k = 0; k < n; ++k)
			if (!dev_scenes_[(size_t) k]) {
				const std::string msg = "GPU " + std::to_string(first + k) + ": " + errors[(size_t) k];
				dev_scenes_.erase(std::remove(dev_scenes_.begin(), dev_scenes_.end(), nullptr), dev_scenes_.end());
				raise(msg);
			}
		kept_root_ = scene.root;
		kept_print_ = print;
		setup_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
	}
	PtxRenderParams rp{};
	rp.width = width_;
	rp.height = height_;
	rp.aa_steps = aa_steps_;
	rp.max_ray_bounces = max_ray_bounces_;
	rp.sample_begin = 0;
	rp.sample_end = aa_steps_ * aa_steps_;
	std::vector<float> color((size_t) width_ * height_ * 3);
	std::vector<unsigned char> rows((size_t) width_ * height_ * 3);
	std::vector<PtxStats> per_gpu(dev_scenes_.size());
	if (ptx_render_image_multi(dev_scenes_.data(), (int32_t) dev_scenes_.size(), &rp, color.data(), rows.data(), &device_seconds_, per_gpu.data()) != PTX_OK)
		raise(ptx_last_error());
	stats_ = per_gpu[0];  // the work of all GPUs together; kernel_ms and kernel_launches stay the first GPU's (the GPUs run side by side)
	for (size_t k = 1; k < per_gpu.size(); ++k) {
		PtxStats const& g = per_gpu[k];
		stats_.samples += g.samples;
		stats_.trace_rays += g.trace_rays;
		stats_.shadow_rays += g.shadow_rays;
		stats_.shaded_hits += g.shaded_hits;
		stats_.launches += g.launches;
		stats_.chunks_redone += g.chunks_redone;
		for (int d = 0; d < 16; ++d) {
			stats_.per_depth_rays[d] += g.per_depth_rays[d];
			stats_.per_depth_shadow[d] += g.per_depth_shadow[d];
			stats_.per_depth_hits[d] += g.per_depth_hits[d];
			for (int j = 0; j < 5; ++j) {
				stats_.trace_tests[d][j] += g.trace_tests[d][j];
				stats_.shadow_tests[d][j] += g.shadow_tests[d][j];
			}
		}
	}
	if (!keep_) drop_scene();
	for (size_t i = 0; i < color_buffer_.size(); ++i) color_buffer_[i] = Color{color[3 * i], color[3 * i + 1], color[3 * i + 2]};
	ppm_.set_rows(rows.data());
	auto end = std::chrono::steady_clock::now();
	std::chrono::duration<double> elapsed_seconds = end - start;
	std::cout << elapsed_seconds.count() << "s rendering\n";
	if (save_ppm_) ppm_.save(filename_);
}

void Renderer::write(Pixel const& p) {  // framework/renderer.cpp:93-106
	size_t buf_pos = (width_ * p.y + p.x);
	if (buf_pos >= color_buffer_.size() || (int) buf_pos < 0) {
		std::cerr << "Fatal Error Renderer::write(Pixel p) : pixel out of ppm_ : " << (int) p.x << "," << (int) p.y << std::endl;
	} else {
		color_buffer_[buf_pos] = p.color;
	}
	ppm_.write(p);
}

float* Renderer::pixel_buffer() const {  // framework/renderer.cpp:293-303: caller owns the array
	auto* pixel_data = new float[(size_t) width_ * height_ * 3];
	for (size_t i = 0; i < color_buffer_.size(); ++i) {
		pixel_data[i * 3] = color_buffer_[i].r;
		pixel_data[i * 3 + 1] = color_buffer_[i].g;
		pixel_data[i * 3 + 2] = color_buffer_[i].b;
	}
	return pixel_data;
}
