#include "renderer.hpp"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>

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
	kept_root_ = nullptr;
}

void Renderer::render(Scene const& scene) {
	auto start = std::chrono::steady_clock::now();
	auto raise = [&](std::string msg) {
		drop_scene();
		error_ = std::move(msg);
		throw static_cast<const char*>(error_.c_str());
	};
	setup_seconds_ = 0;
	if (!(keep_ && kept_root_ == scene.root.get() && !dev_scenes_.empty())) {
		drop_scene();
		FlatScene flat;
		try {
			flat = flatten_scene(scene);
		} catch (SceneError const& e) {
			raise(e.message);
		}
		const int visible = ptx_device_count();
		if (visible < 1) raise("no gfx950 GPU: this Renderer has no CPU path");
		int n = n_devices_ <= 0 ? visible - first_device_ : n_devices_;
		if (first_device_ < 0 || first_device_ >= visible || n < 1 || first_device_ + n > visible) raise("the GPUs asked for are not there");
		// every GPU renders at least one sample of every pixel
		n = (int) std::min<unsigned long long>((unsigned long long) n, (unsigned long long) aa_steps_ * aa_steps_);
		for (int k = 0; k < n; ++k) {
			PtxScene* s = nullptr;
			if (ptx_scene_create(&flat.desc, first_device_ + k, 0, &s) != PTX_OK) raise(ptx_last_error());
			dev_scenes_.push_back(s);
		}
		kept_root_ = scene.root.get();
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
	stats_ = per_gpu[0];  // the work of all GPUs together (times: the first GPU's)
	for (size_t k = 1; k < per_gpu.size(); ++k) {
		stats_.samples += per_gpu[k].samples;
		stats_.trace_rays += per_gpu[k].trace_rays;
		stats_.shadow_rays += per_gpu[k].shadow_rays;
		stats_.shaded_hits += per_gpu[k].shaded_hits;
		stats_.launches += per_gpu[k].launches;
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
