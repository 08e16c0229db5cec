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
void PpmWriter::save() {  // framework/ppmwriter.cpp:55-76
	std::string text = "P3 " + std::to_string((unsigned int) width_) + " " + std::to_string((unsigned int) height_) + " 255 \n";
	text.reserve(data_.size() * 4 + 64);
	std::size_t pos = 1;
	for (auto it = data_.begin(); it < data_.end(); ++it, ++pos) {
		text += std::to_string((int) *it);
		text += ' ';
		if (pos > 18) {
			text += '\n';
			pos = 0;
		}
	}
	std::ofstream out(file_.c_str(), std::ios::out | std::ios::binary);
	out.write(text.data(), (std::streamsize) text.size());
}

// ---------------------------------------------------------------- Renderer
Renderer::Renderer(unsigned w, unsigned h, std::string const& file, unsigned aa_steps, unsigned max_ray_bounces)
    : width_(w), height_(h), color_buffer_(w * h, Color{0.0, 0.0, 0.0}), filename_(file), ppm_(width_, height_), aa_steps_(aa_steps),
      max_ray_bounces_(max_ray_bounces) {}

void Renderer::render(Scene const& scene) {
	auto start = std::chrono::steady_clock::now();
	PtxScene* dev_scene = nullptr;
	auto raise = [&](std::string msg) {
		if (dev_scene) ptx_scene_destroy(dev_scene);
		error_ = std::move(msg);
		throw static_cast<const char*>(error_.c_str());
	};
	FlatScene flat;
	try {
		flat = flatten_scene(scene);
	} catch (SceneError const& e) {
		raise(e.message);
	}
	if (ptx_scene_create(&flat.desc, device_, 0, &dev_scene) != PTX_OK) raise(ptx_last_error());
	PtxRenderParams rp{};
	rp.width = width_;
	rp.height = height_;
	rp.aa_steps = aa_steps_;
	rp.max_ray_bounces = max_ray_bounces_;
	rp.sample_begin = 0;
	rp.sample_end = aa_steps_ * aa_steps_;
	std::vector<float> color((size_t) width_ * height_ * 3);
	std::vector<unsigned char> rows((size_t) width_ * height_ * 3);
	if (ptx_render_image(dev_scene, &rp, color.data(), rows.data(), &device_seconds_, &stats_) != PTX_OK) raise(ptx_last_error());
	ptx_scene_destroy(dev_scene);
	dev_scene = nullptr;
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
