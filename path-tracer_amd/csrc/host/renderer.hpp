// Renderer with the reference's interface (framework/renderer.hpp:22-62).  render() hands
// the scene to the gfx950 library behind include/ptx.h instead of spawning CPU threads.
#pragma once
#include <string>
#include <vector>

#include "ppmwriter.hpp"
#include "ptx.h"
#include "scene.hpp"

class Renderer {
public:
	Renderer(unsigned w, unsigned h, std::string const& file, unsigned aa_steps, unsigned max_ray_bounces);

	// Blocking, like the reference.  Fills color_buffer(), writes the PPM file and prints
	// "<seconds>s rendering" (framework/renderer.cpp:56).  Throws `const char*` on failure,
	// the reference's error convention (source/example.cpp:25-27).
	void render(Scene const& scene);
	void write(Pixel const& p);

	inline std::vector<Color> const& color_buffer() const { return color_buffer_; }
	[[nodiscard]] float* pixel_buffer() const;

	// additions (no counterpart in the reference)
	void set_device(int device) { device_ = device; }
	void set_output(bool save_ppm) { save_ppm_ = save_ppm; }
	PtxStats const& stats() const { return stats_; }
	double device_seconds() const { return device_seconds_; }

private:
	unsigned width_;
	unsigned height_;
	std::vector<Color> color_buffer_;
	std::string filename_;
	PpmWriter ppm_;
	unsigned aa_steps_;
	unsigned max_ray_bounces_;
	int device_ = 0;
	bool save_ppm_ = true;
	PtxStats stats_{};
	double device_seconds_ = 0;
	std::string error_;
};
