// extern "C" helpers over the host-side scene model, for callers without a C++ compiler in
// the loop (the Python tests and bench.py load scenes through these).  The render path
// itself is include/ptx.h; nothing here computes pixels.
#include <cstring>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "renderer.hpp"

namespace {
struct HostScene {
	Scene scene;
	FlatScene flat;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* pth_last_error(void) { return g_err.c_str(); }

// load_scene + flatten_scene.  obj_dir may be NULL (folder of the scene file).
void* pth_load_scene(const char* sdf_path, const char* obj_dir) {
	try {
		auto* h = new HostScene{obj_dir ? load_scene(sdf_path, obj_dir) : load_scene(sdf_path), FlatScene{}};
		try {
			h->flat = flatten_scene(h->scene);
		} catch (...) {
			delete h;
			throw;
		}
		return h;
	} catch (SceneError const& e) {
		g_err = e.message;
	} catch (const char* msg) {
		g_err = msg;
	} catch (std::exception const& e) {
		g_err = e.what();
	}
	return nullptr;
}
void pth_free_scene(void* h) { delete static_cast<HostScene*>(h); }

// What a C++ caller of the reference does to move several shapes together: a Composite of its own,
//   auto g = std::make_shared<Composite>(name); g->add_child(std::make_shared<Sphere>(r, c, n, mat)); ...
//   g->build_octree(); scene.root->add_child(g); scene.root->build_octree();
// (framework/composite.hpp:9-29, sphere.hpp:10, box.hpp:9).  n children: kinds[i] 0 = sphere (params cx cy cz r),
// 1 = box (params min xyz, max xyz), 7 floats per child; names and material names joined by ';'.
int pth_scene_add_group(void* h, const char* name, int n, const int* kinds, const float* params, const char* child_names, const char* material_names) {
	auto* hs = static_cast<HostScene*>(h);
	try {
		auto split = [](std::string s) {
			std::vector<std::string> out;
			size_t cut;
			while ((cut = s.find(';')) != std::string::npos) {
				out.push_back(s.substr(0, cut));
				s = s.substr(cut + 1);
			}
			out.push_back(s);
			return out;
		};
		const std::vector<std::string> names = split(child_names), mats = split(material_names);
		if ((int) names.size() != n || (int) mats.size() != n) throw SceneError{"add_group: n names and n materials expected"};
		// "a/b/name": the new Composite goes into Composite b inside Composite a under the root
		std::vector<std::shared_ptr<Composite>> above{hs->scene.root};
		std::string own(name ? name : "");
		for (size_t cut; (cut = own.find('/')) != std::string::npos; own = own.substr(cut + 1)) {
			std::shared_ptr<Shape> next = above.back()->find_child(own.substr(0, cut));
			if (next->kind() != Shape::Kind::composite) throw SceneError{"'" + next->get_name() + "' has no children"};
			above.push_back(std::static_pointer_cast<Composite>(next));
		}
		auto g = std::make_shared<Composite>(own);
		for (int i = 0; i < n; ++i) {
			const float* p = params + 7 * i;
			std::shared_ptr<Material> m = hs->scene.find_mat(mats[(size_t) i]);
			if (kinds[i] == 0) g->add_child(std::make_shared<Sphere>(p[3], vec3{p[0], p[1], p[2]}, names[(size_t) i], m));
			else g->add_child(std::make_shared<Box>(vec3{p[0], p[1], p[2]}, vec3{p[3], p[4], p[5]}, names[(size_t) i], m));
		}
		g->build_octree();
		above.back()->add_child(g);
		for (size_t k = above.size(); k-- > 0;) above[k]->build_octree();  // (bounds of everything above, innermost first)
		hs->flat = flatten_scene(hs->scene);
		return 0;
	} catch (SceneError const& e) {
		g_err = e.message;
	} catch (const char* msg) {
		g_err = msg;
	} catch (std::exception const& e) {
		g_err = e.what();
	}
	return 1;
}

// What a C++ caller of the reference does after load_scene to move a part of a loaded mesh:
// scene.root->find_child(a)->find_child(b)...->translate / rotate / scale (framework/composite.cpp:112-114,
// :168-181; shape.cpp:17-36).  `path` names the child, levels separated by '/'; op 1 = translate(a, b, c),
// 2 = rotate(yaw a, pitch b, roll c; radians), 3 = scale(a, b, c).  The scene is flattened again.
int pth_scene_transform_child(void* h, const char* path, int op, float a, float b, float c) {
	auto* hs = static_cast<HostScene*>(h);
	try {
		std::shared_ptr<Shape> shape = hs->scene.root;
		std::string rest(path ? path : "");
		while (!rest.empty()) {
			const size_t cut = rest.find('/');
			const std::string name = rest.substr(0, cut);
			rest = cut == std::string::npos ? std::string() : rest.substr(cut + 1);
			if (shape->kind() != Shape::Kind::composite) throw SceneError{"'" + shape->get_name() + "' has no children"};
			shape = static_cast<Composite&>(*shape).find_child(name);
		}
		if (op == 1) shape->translate(a, b, c);
		else if (op == 2) shape->rotate(a, b, c);
		else if (op == 3) shape->scale(a, b, c);
		else throw SceneError{"unknown transform"};
		hs->flat = flatten_scene(hs->scene);
		return 0;
	} catch (SceneError const& e) {
		g_err = e.message;
	} catch (std::exception const& e) {
		g_err = e.what();
	}
	return 1;
}

const PtxSceneDesc* pth_scene_desc(void* h) { return &static_cast<HostScene*>(h)->flat.desc; }
const char* pth_prim_name(void* h, int i) { return static_cast<HostScene*>(h)->flat.prim_names[(size_t) i].c_str(); }
const char* pth_prim_path(void* h, int i) { return static_cast<HostScene*>(h)->flat.prim_paths[(size_t) i].c_str(); }
void pth_tree_stats(void* h, int* composites, int* leaf_entries, int* max_depth) {
	FlatScene const& f = static_cast<HostScene*>(h)->flat;
	*composites = f.composites;
	*leaf_entries = f.leaf_entries;
	*max_depth = f.max_tree_depth;
}

// World matrix and inverse (16 floats each, column-major) of a Shape that got the transforms
// `xf` in order, through the mirror's Shape::translate / rotate / scale (framework/shape.cpp:17-36):
// n_xf ops of 4 floats {op, a, b, c}, op 1 = translate(a, b, c), 2 = rotate(yaw a, pitch b, roll c;
// radians, the unit of Shape::rotate), 3 = scale(a, b, c).  For callers that describe single shapes
// themselves (PtxPrim::world / world_inv) instead of loading a scene file.
void pth_shape_matrices(const float* xf, int n_xf, float* world16, float* world_inv16) {
	Sphere sh;
	for (int k = 0; k < n_xf; ++k) {
		const float* q = xf + 4 * k;
		if (q[0] == 1) sh.translate(q[1], q[2], q[3]);
		else if (q[0] == 2) sh.rotate(q[1], q[2], q[3]);
		else if (q[0] == 3) sh.scale(q[1], q[2], q[3]);
	}
	std::memcpy(world16, &sh.world_transform(), 16 * sizeof(float));
	std::memcpy(world_inv16, &sh.world_transform_inv(), 16 * sizeof(float));
}

// The drop-in call sequence of source/example.cpp:11-21: Renderer{w,h,file,aa,bounces}.render(scene).
// color may be NULL.  Returns 0, or -1 with pth_last_error() set.
int pth_render_file(void* h, unsigned w, unsigned hgt, unsigned aa, unsigned bounces, const char* ppm_path, float* color, double* device_seconds) {
	try {
		Renderer r{w, hgt, ppm_path ? ppm_path : "", aa, bounces};
		r.set_output(ppm_path != nullptr);
		r.render(static_cast<HostScene*>(h)->scene);
		if (color) {
			auto const& cb = r.color_buffer();
			for (size_t i = 0; i < cb.size(); ++i) {
				color[3 * i] = cb[i].r;
				color[3 * i + 1] = cb[i].g;
				color[3 * i + 2] = cb[i].b;
			}
		}
		if (device_seconds) *device_seconds = r.device_seconds();
		return 0;
	} catch (const char* msg) {
		g_err = msg;
	} catch (SceneError const& e) {
		g_err = e.message;
	}
	return -1;
}

// The same through a Renderer object that lives across calls (keep_scene, GPU selection).
void* pth_renderer_create(unsigned w, unsigned hgt, unsigned aa, unsigned bounces, const char* ppm_path) {
	auto* r = new Renderer{w, hgt, ppm_path ? ppm_path : "", aa, bounces};
	r->set_output(ppm_path != nullptr);
	return r;
}
void pth_renderer_destroy(void* r) { delete static_cast<Renderer*>(r); }
// first_device < 0: leave the GPU selection alone (one GPU, the first); n_devices 0 = all from first_device on
void pth_renderer_options(void* r, int first_device, int n_devices, int keep_scene) {
	auto* R = static_cast<Renderer*>(r);
	if (first_device >= 0) R->set_devices(first_device, n_devices);
	R->keep_scene(keep_scene != 0);
}
// Renderer::select_devices: the GPUs a selection comes to (no GPU needed to ask)
int pth_select_devices(int first_device, int n_devices, int visible, unsigned long long samples, int* out_first, int* out_count) {
	return Renderer::select_devices(first_device, n_devices, visible, samples, out_first, out_count) ? 0 : -1;
}
int pth_renderer_render(void* r, void* h, float* color, double* device_seconds, double* setup_seconds) {
	auto* R = static_cast<Renderer*>(r);
	try {
		R->render(static_cast<HostScene*>(h)->scene);
		if (color) {
			auto const& cb = R->color_buffer();
			for (size_t i = 0; i < cb.size(); ++i) {
				color[3 * i] = cb[i].r;
				color[3 * i + 1] = cb[i].g;
				color[3 * i + 2] = cb[i].b;
			}
		}
		if (device_seconds) *device_seconds = R->device_seconds();
		if (setup_seconds) *setup_seconds = R->setup_seconds();
		return 0;
	} catch (const char* msg) {
		g_err = msg;
	} catch (SceneError const& e) {
		g_err = e.message;
	}
	return -1;
}

}  // extern "C"
