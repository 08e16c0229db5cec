// PpmWriter with the reference's interface and byte-identical output
// (framework/ppmwriter.hpp:19-38, ppmwriter.cpp:18-76).
#pragma once
#include <string>
#include <vector>

#include "scene.hpp"

class PpmWriter {
public:
	PpmWriter(std::size_t w, std::size_t h, std::string const& file) : file_(file), data_(w * h * 3), width_(w), height_(h) {}
	PpmWriter(std::size_t w, std::size_t h) : PpmWriter(w, h, "untitled.ppm") {}

	void write(Pixel const& p);
	void save(std::string const& file);
	void save();

	// rows already flipped and quantised (what the GPU resolve produces): w*h*3 bytes
	void set_rows(const unsigned char* rgb8);
	std::vector<unsigned int> const& data() const { return data_; }

private:
	std::string file_;
	std::vector<unsigned int> data_;
	std::size_t width_;
	std::size_t height_;
};
