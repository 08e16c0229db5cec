// Renderer with the reference's interface (framework/renderer.hpp:22-62).  render() hands
// the scene to the gfx950 library behind include/ptx.h instead of spawning CPU threads.
#pragma once
#include <memory>
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

	~Renderer();
	Renderer(Renderer const&) = delete;
	Renderer& operator=(Renderer const&) = delete;

	// additions (no counterpart in the reference)
	// Which GPUs render() uses.  The reference spreads an image over every hardware thread
	// (framework/renderer.cpp:39); here the default is ONE GPU (the first), and spreading an image
	// over several - each taking a slice of the per-pixel samples, ptx_render_image_multi - is asked
	// for: set_device(d) = GPU d only; set_devices(first, n) = n GPUs from `first` on (n = 0: all from
	// `first` on); set_devices(n) = the first n.
	void set_device(int device) { set_devices(device, 1); }
	void set_devices(int n) { set_devices(0, n); }
	void set_devices(int first, int n) { first_device_ = first; n_devices_ = n; drop_scene(); }
	// The GPUs a selection comes to: [*first, *first + *count) of `visible` GPUs for an image of `samples`
	// samples per pixel (every GPU renders at least one); false when the GPUs asked for are not there.
	static bool select_devices(int first, int n, int visible, unsigned long long samples, int* out_first, int* count);
	// keep = true: the flattened scene stays on the GPUs after render(), and the next render() of
	// the same Scene - the same root object, still alive, with the same camera, lights, ambient term
	// and materials - skips flattening, BVH build and upload.  A change to the shapes below the root is
	// not seen: the caller calls drop_scene() after one.
	void keep_scene(bool keep) { keep_ = keep; if (!keep) drop_scene(); }
	void drop_scene();
	void set_output(bool save_ppm) { save_ppm_ = save_ppm; }
	double setup_seconds() const { return setup_seconds_; }  // flatten + upload part of the last render() (0 when the kept scene was used)
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
	int first_device_ = 0, n_devices_ = 1;
	bool keep_ = false;
	std::weak_ptr<Composite> kept_root_;   // the root of the scene that is on the GPUs (expired: that scene is gone)
	unsigned long long kept_print_ = 0;    // and a fingerprint of everything of it that lives outside the shape tree
	std::vector<PtxScene*> dev_scenes_;
	double setup_seconds_ = 0;
	bool save_ppm_ = true;
	PtxStats stats_{};
	double device_seconds_ = 0;
	std::string error_;
};
