// TEST INFRASTRUCTURE ONLY. Thin extern "C" shim that is compiled TOGETHER WITH the
// unmodified reference sources where they lie under /root/reference (see oracle/Makefile,
// target _ref/libref.so). Nothing of the reference is copied into this repository; this
// file only calls the reference's public API:
//   load_scene            framework/scene.hpp:26
//   Renderer{..}.render   framework/renderer.hpp:24-26
//   Shape::intersect      framework/shape.hpp:28   (through scene.root)
// It is used (a) to pin oracle/cpu_ref.cpp bit-for-bit, (b) to generate tests/golden/*,
// (c) optionally as the "reference" CPU baseline of bench.py.  The product never loads it.
#include <cstring>
#include <cstdio>
#include <chrono>
#include <string>
#include <vector>
#include <unistd.h>
#include <limits.h>
#include <sys/stat.h>

#include "scene.hpp"
#include "renderer.hpp"
#include "sphere.hpp"
#include "box.hpp"
#include "triangle.hpp"

namespace {
// The reference opens OBJ files at the literal path "../../sdf/<name>.obj"
// (framework/scene.cpp:326), so the process must sit two levels below the folder that
// holds sdf/.  <dir of sdf>/../.refcwd/x is created and entered for the load.
std::string enter_scene_cwd(const char* sdf_path, std::string& file_name) {
	char real[PATH_MAX];
	if (!realpath(sdf_path, real)) return "";
	std::string p(real);
	auto slash = p.find_last_of('/');
	std::string dir = p.substr(0, slash);          // .../sdf
	file_name = p.substr(slash + 1);
	std::string base = dir.substr(0, dir.find_last_of('/'));
	std::string c1 = base + "/.refcwd";
	std::string c2 = c1 + "/x";
	mkdir(c1.c_str(), 0755);
	mkdir(c2.c_str(), 0755);
	char old[PATH_MAX];
	if (!getcwd(old, sizeof old)) return "";
	if (chdir(c2.c_str()) != 0) return "";
	// the folder that holds the scene must be called "sdf" for the OBJ path to resolve
	return std::string(old);
}
}

extern "C" {

void* ref_load_scene(const char* sdf_path) {
	std::string file_name;
	std::string old = enter_scene_cwd(sdf_path, file_name);
	if (old.empty()) return nullptr;
	// path is given relative to the new cwd through the real directory name
	char real[PATH_MAX];
	std::string rel;
	{
		// <cwd>/../../<dirname>/<file>
		std::string p(sdf_path);
		if (!realpath(sdf_path, real)) { real[0] = 0; }
		rel = real[0] ? std::string(real) : p;
	}
	Scene* scene = new Scene(load_scene(rel));
	if (chdir(old.c_str()) != 0) { /* keep going */ }
	return scene;
}

void ref_free_scene(void* s) { delete static_cast<Scene*>(s); }

// Renders with the reference renderer; returns wall seconds of Renderer::render (which
// includes the PPM text write, framework/renderer.cpp:58).
double ref_render(void* s, unsigned w, unsigned h, unsigned aa, unsigned bounces,
                  const char* ppm_out, float* color_buffer /* w*h*3 or null */) {
	Scene* scene = static_cast<Scene*>(s);
	Renderer r{w, h, std::string(ppm_out), aa, bounces};
	auto t0 = std::chrono::steady_clock::now();
	r.render(*scene);
	auto t1 = std::chrono::steady_clock::now();
	if (color_buffer) {
		auto const& cb = r.color_buffer();
		for (size_t i = 0; i < cb.size(); ++i) {
			color_buffer[3 * i + 0] = cb[i].r;
			color_buffer[3 * i + 1] = cb[i].g;
			color_buffer[3 * i + 2] = cb[i].b;
		}
	}
	return std::chrono::duration<double>(t1 - t0).count();
}

// Primary rays exactly as framework/renderer.cpp:27-37 and :76-82 build them
// (sample (aax, aay) of every pixel, pixel-major, aax outer).
void ref_primary_rays(void* s, unsigned w, unsigned h, unsigned aa, float* origins, float* dirs) {
	Scene* scene = static_cast<Scene*>(s);
	Camera cam = scene->camera;
	float fov_radians = cam.fov_x / 180 * 3.14159265f;
	float img_plane_dist = (w / 2.0f) / tan(fov_radians / 2);
	glm::vec3 u = glm::cross(cam.direction, cam.up);
	glm::vec3 v = glm::cross(u, cam.direction);
	glm::mat4 trans_mat{glm::vec4{u, 0}, glm::vec4{v, 0}, glm::vec4{-cam.direction, 0}, glm::vec4{cam.position, 1}};
	float aa_unit = 1.0f / aa;
	size_t k = 0;
	for (unsigned y = 0; y < h; ++y)
		for (unsigned x = 0; x < w; ++x)
			for (int aax = 0; aax < (int) aa; ++aax)
				for (int aay = 0; aay < (int) aa; ++aay) {
					glm::vec3 pixel_pos = glm::vec3{x + aax * aa_unit - (w * 0.5f), y + aay * aa_unit - (h * 0.5f), -img_plane_dist};
					glm::vec4 d = trans_mat * glm::vec4{glm::normalize(pixel_pos), 0};
					glm::vec3 o = glm::vec3{trans_mat[3]};
					origins[3 * k + 0] = o.x; origins[3 * k + 1] = o.y; origins[3 * k + 2] = o.z;
					dirs[3 * k + 0] = d.x; dirs[3 * k + 1] = d.y; dirs[3 * k + 2] = d.z;
					++k;
				}
}

// scene.root->intersect(ray) for n rays. names: n records of 48 bytes (NUL padded).
void ref_intersect(void* s, long n, const float* o, const float* d, int* hit, float* t,
                   float* pos, float* nrm, float* rdir, char* names) {
	Scene* scene = static_cast<Scene*>(s);
	for (long i = 0; i < n; ++i) {
		Ray ray{glm::vec3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, glm::vec3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
		HitPoint hp = scene->root->intersect(ray);
		hit[i] = hp.does_intersect ? 1 : 0;
		t[i] = hp.distance;
		for (int c = 0; c < 3; ++c) {
			pos[3 * i + c] = hp.position[c];
			nrm[3 * i + c] = hp.surface_normal[c];
			rdir[3 * i + c] = hp.ray_direction[c];
		}
		if (names) {
			std::memset(names + 48 * i, 0, 48);
			std::strncpy(names + 48 * i, hp.hit_object.c_str(), 47);
		}
	}
}

// Per-primitive known-answer probes, constructed the way source/tests.cpp does
// (tests.cpp:76-96 sphere, :112-142 box, :144-158 triangle, :252-259 scaled sphere).
// kind 0 sphere(c, r) 1 box(min,max) 2 triangle(v0,v1,v2); p = 9 floats of parameters;
// xf = up to 3 ops {op(0 none,1 translate,2 rotate(rad),3 scale), a, b, c}.
void ref_prim_intersect(int kind, const float* p, const float* xf, int nxf, long n,
                        const float* o, const float* d, int* hit, float* t, float* pos,
                        float* nrm, float* rdir) {
	std::shared_ptr<Material> mat = std::make_shared<Material>();
	std::shared_ptr<Shape> sh;
	if (kind == 0) sh = std::make_shared<Sphere>(p[3], glm::vec3{p[0], p[1], p[2]}, "s", mat);
	else if (kind == 1) sh = std::make_shared<Box>(glm::vec3{p[0], p[1], p[2]}, glm::vec3{p[3], p[4], p[5]}, "b", mat);
	else sh = std::make_shared<Triangle>(glm::vec3{p[0], p[1], p[2]}, glm::vec3{p[3], p[4], p[5]}, glm::vec3{p[6], p[7], p[8]}, "t", mat);
	for (int k = 0; k < nxf; ++k) {
		const float* q = xf + 4 * k;
		if (q[0] == 1) sh->translate(q[1], q[2], q[3]);
		else if (q[0] == 2) sh->rotate(q[1], q[2], q[3]);
		else if (q[0] == 3) sh->scale(q[1], q[2], q[3]);
	}
	for (long i = 0; i < n; ++i) {
		Ray ray{glm::vec3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, glm::vec3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
		HitPoint hp = sh->intersect(ray);
		hit[i] = hp.does_intersect ? 1 : 0;
		t[i] = hp.distance;
		for (int c = 0; c < 3; ++c) {
			pos[3 * i + c] = hp.position[c];
			nrm[3 * i + c] = hp.surface_normal[c];
			rdir[3 * i + c] = hp.ray_direction[c];
		}
	}
}

// camera (fov, pos, dir, up), ambient intensity, lights (pos, intensity) for cross-checks
void ref_scene_info(void* s, float* cam10, float* ambient3, int* n_lights, float* lights /* 6 each, max 16 */,
                    float* root_min3, float* root_max3) {
	Scene* scene = static_cast<Scene*>(s);
	cam10[0] = scene->camera.fov_x;
	for (int c = 0; c < 3; ++c) {
		cam10[1 + c] = scene->camera.position[c];
		cam10[4 + c] = scene->camera.direction[c];
		cam10[7 + c] = scene->camera.up[c];
	}
	ambient3[0] = scene->ambient.intensity.r; ambient3[1] = scene->ambient.intensity.g; ambient3[2] = scene->ambient.intensity.b;
	*n_lights = (int) scene->lights.size();
	for (int i = 0; i < *n_lights && i < 16; ++i) {
		for (int c = 0; c < 3; ++c) lights[6 * i + c] = scene->lights[i].position[c];
		lights[6 * i + 3] = scene->lights[i].intensity.r;
		lights[6 * i + 4] = scene->lights[i].intensity.g;
		lights[6 * i + 5] = scene->lights[i].intensity.b;
	}
	glm::vec3 mn = scene->root->min(), mx = scene->root->max();
	for (int c = 0; c < 3; ++c) { root_min3[c] = mn[c]; root_max3[c] = mx[c]; }
}

}
