// TEST INFRASTRUCTURE ONLY (container-only: compiled together with the unmodified reference
// sources into _ref/libref.so, see oracle/Makefile).  The scene hand-over (flattening a Scene into the arrays of include/ptx.h)
// written against the REFERENCE's own classes (glm types, std::map children, shared_ptr
// materials): walks a Scene loaded by the reference's load_scene and fills the PtxSceneDesc
// arrays of include/ptx.h.  oracle/check_flatten.py compares the result byte for byte with
// what the product's host mirror (libpt_host.so) produces for the same scene file, which is
// the proof that a maintainer who adds the accessors and this walk to the reference feeds
// the library the same bytes the tests feed it.
// The reference classes keep their members private and have no accessors
// (framework/shape.hpp:36-41, composite.hpp:34-35, sphere.hpp:22-23, box.hpp:27-28,
// triangle.hpp:32-35); this probe opens them with the preprocessor instead of editing the files.
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#define private public
#define protected public
#include "scene.hpp"
#include "sphere.hpp"
#include "box.hpp"
#include "triangle.hpp"
#include "composite.hpp"
#undef private
#undef protected

#include <cstring>
#include <unordered_map>

#include "ptx.h"

namespace {

bool is_identity(glm::mat4 const& m) {
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r)
			if (m[c][r] != (c == r ? 1.0f : 0.0f)) return false;
	return true;
}
Composite const* as_composite(Shape const& s) { return dynamic_cast<Composite const*>(&s); }

bool only_plain_triangles(Composite const& c) {
	for (auto const& kv : c.children_) {
		Shape const& s = *kv.second;
		if (!is_identity(s.world_transform_)) return false;
		if (Composite const* sub = as_composite(s)) {
			if (!only_plain_triangles(*sub)) return false;
		} else if (!dynamic_cast<Triangle const*>(&s)) {
			return false;
		}
	}
	return !c.children_.empty();
}

// an untransformed Composite that holds a sphere, a box or a moved triangle (not below a transformed Composite): handed
// over as a frame of its own, so that its bounds test is kept
bool holds_other_leaves(Composite const& c) {
	for (auto const& kv : c.children_) {
		Shape const& s = *kv.second;
		if (Composite const* sub = as_composite(s)) {
			if (is_identity(s.world_transform_) && holds_other_leaves(*sub)) return true;
		} else if (!dynamic_cast<Triangle const*>(&s) || !is_identity(s.world_transform_)) {
			return true;
		}
	}
	return false;
}

struct Flat {
	std::vector<PtxFrame> frames;
	std::vector<PtxPrim> prims;
	std::vector<PtxMaterial> materials;
	std::vector<PtxLight> lights;
	std::vector<PtxGuard> guards;
	std::vector<PtxOccurrence> occurrences;
	std::vector<std::string> names;
	float ambient[3];
	PtxCamera camera;
	std::string error;
};

struct Walker {
	Flat& out;
	std::unordered_map<Shape const*, int> seen;
	std::unordered_map<Material const*, int> mat_index;

	int material_of(Shape const& s) {
		Material const* m = s.material_.get();
		if (!m) throw std::string("shape '" + s.name_ + "' has no material");
		auto it = mat_index.find(m);
		if (it != mat_index.end()) return it->second;
		PtxMaterial pm{};
		pm.ka[0] = m->ka.r; pm.ka[1] = m->ka.g; pm.ka[2] = m->ka.b;
		pm.kd[0] = m->kd.r; pm.kd[1] = m->kd.g; pm.kd[2] = m->kd.b;
		pm.ks[0] = m->ks.r; pm.ks[1] = m->ks.g; pm.ks[2] = m->ks.b;
		pm.m = m->m;
		pm.glossy = m->glossy;
		pm.opacity = m->opacity;
		pm.ior = m->ior;
		int idx = (int) out.materials.size();
		out.materials.push_back(pm);
		mat_index.emplace(m, idx);
		return idx;
	}
	void emit_leaf(Shape const& s, int frame, int levels_in) {
		if (!seen.emplace(&s, (int) out.prims.size()).second) return;
		PtxPrim p{};
		p.frame = frame;
		p.levels_in = levels_in;
		p.material = material_of(s);
		std::memcpy(p.world, &s.world_transform_[0][0], sizeof p.world);          // a glm::mat4 is 16 column-major floats
		std::memcpy(p.world_inv, &s.world_transform_inv_[0][0], sizeof p.world_inv);
		if (auto const* q = dynamic_cast<Sphere const*>(&s)) {
			p.kind = PTX_SPHERE;
			p.p[0] = q->center_.x; p.p[1] = q->center_.y; p.p[2] = q->center_.z; p.p[3] = q->radius_;
		} else if (auto const* q = dynamic_cast<Box const*>(&s)) {
			p.kind = PTX_BOX;
			p.p[0] = q->min_.x; p.p[1] = q->min_.y; p.p[2] = q->min_.z;
			p.p[3] = q->max_.x; p.p[4] = q->max_.y; p.p[5] = q->max_.z;
		} else if (auto const* q = dynamic_cast<Triangle const*>(&s)) {
			p.kind = PTX_TRIANGLE;
			p.p[0] = q->v0_.x; p.p[1] = q->v0_.y; p.p[2] = q->v0_.z;
			p.p[3] = q->v1_.x; p.p[4] = q->v1_.y; p.p[5] = q->v1_.z;
			p.p[6] = q->v2_.x; p.p[7] = q->v2_.y; p.p[8] = q->v2_.z;
			p.normal[0] = q->n_.x; p.normal[1] = q->n_.y; p.normal[2] = q->n_.z;
		} else {
			throw std::string("unexpected shape kind");
		}
		out.prims.push_back(p);
		out.names.push_back(s.name_);
	}
	void frame_of(Composite const& c, int parent, int levels_up, PtxFrame& f) {
		std::memcpy(f.world, &c.world_transform_[0][0], sizeof f.world);
		std::memcpy(f.world_inv, &c.world_transform_inv_[0][0], sizeof f.world_inv);
		glm::vec3 lo = c.bounds_->min_, hi = c.bounds_->max_;
		f.bounds_min[0] = lo.x; f.bounds_min[1] = lo.y; f.bounds_min[2] = lo.z;
		f.bounds_max[0] = hi.x; f.bounds_max[1] = hi.y; f.bounds_max[2] = hi.z;
		f.parent = parent;
		f.levels_up = levels_up;
	}
	void walk_mesh(Composite const& c, int frame, int levels) {
		for (auto const& kv : c.children_) {
			Shape const& s = *kv.second;
			if (Composite const* sub = as_composite(s)) {
				if (!is_identity(s.world_transform_) || holds_other_leaves(*sub)) {  // a transformed Composite inside a transformed one (or one that must keep its own bounds test): a frame inside this frame
					if (!seen.emplace(&s, -(int) out.frames.size()).second) continue;
					if (!sub->bounds_) throw std::string("Composite without bounds");
					PtxFrame f{};
					frame_of(*sub, frame, levels, f);
					out.frames.push_back(f);
					walk_mesh(*sub, (int) out.frames.size() - 1, 0);
					continue;
				}
				walk_mesh(*sub, frame, levels + 1);
			} else {
				emit_leaf(s, frame, levels);
			}
		}
	}
	// c: the root (guard -1) or one of its octants (guard = its index in out.guards)
	void walk_root(Composite const& c, int levels, int guard) {
		for (auto const& kv : c.children_) {
			Shape const& s = *kv.second;
			Composite const* sub = as_composite(s);
			if (!sub) {
				emit_leaf(s, 0, levels);
				if (out.prims[(size_t) seen.at(&s)].frame == 0) out.occurrences.push_back(PtxOccurrence{guard, seen.at(&s), 0, 0});
				continue;
			}
			if (is_identity(sub->world_transform_) && !only_plain_triangles(*sub)) {
				if (!sub->bounds_) throw std::string("Composite without bounds");
				PtxGuard g{};
				glm::vec3 lo = sub->bounds_->min_, hi = sub->bounds_->max_;
				g.bounds_min[0] = lo.x; g.bounds_min[1] = lo.y; g.bounds_min[2] = lo.z;
				g.bounds_max[0] = hi.x; g.bounds_max[1] = hi.y; g.bounds_max[2] = hi.z;
				g.parent = guard;
				g.levels = levels + 1;
				int idx = (int) out.guards.size();
				out.guards.push_back(g);
				walk_root(*sub, levels + 1, idx);
				continue;
			}
			auto ins = seen.emplace(&s, -(int) out.frames.size());
			if (!ins.second) {
				out.occurrences.push_back(PtxOccurrence{guard, -1, -ins.first->second, 0});
				continue;
			}
			if (!sub->bounds_) throw std::string("Composite without bounds");
			PtxFrame f{};
			frame_of(*sub, 0, levels, f);
			int idx = (int) out.frames.size();
			out.frames.push_back(f);
			out.occurrences.push_back(PtxOccurrence{guard, -1, idx, 0});
			walk_mesh(*sub, idx, 0);
		}
	}
};

}  // namespace

extern "C" {

// Flattens a Scene* obtained from ref_load_scene.  Returns a handle (free with ref_flat_free)
// or NULL; the getters hand out the arrays and their element counts.
void* ref_flatten(void* scene_ptr) {
	Scene* scene = static_cast<Scene*>(scene_ptr);
	auto* out = new Flat{};
	try {
		Composite const& root = *scene->root;
		if (!is_identity(root.world_transform_) || !root.bounds_) throw std::string("root transformed or without bounds");
		PtxFrame f0{};
		Walker w{*out, {}, {}};
		w.frame_of(root, -1, 0, f0);
		out->frames.push_back(f0);
		w.walk_root(root, 1, -1);
		if (out->guards.empty()) out->occurrences.clear();
		for (PointLight const& L : scene->lights) {
			PtxLight pl{};
			pl.position[0] = L.position.x; pl.position[1] = L.position.y; pl.position[2] = L.position.z;
			pl.intensity[0] = L.intensity.r; pl.intensity[1] = L.intensity.g; pl.intensity[2] = L.intensity.b;
			out->lights.push_back(pl);
		}
		out->ambient[0] = scene->ambient.intensity.r; out->ambient[1] = scene->ambient.intensity.g; out->ambient[2] = scene->ambient.intensity.b;
		out->camera.fov_x = scene->camera.fov_x;
		for (int k = 0; k < 3; ++k) {
			out->camera.position[k] = scene->camera.position[k];
			out->camera.direction[k] = scene->camera.direction[k];
			out->camera.up[k] = scene->camera.up[k];
		}
	} catch (std::string const& e) {
		out->error = e;
	}
	return out;
}
// A Composite of the caller's own with spheres and boxes in it, built with the reference's constructors and added to
// the root (composite.hpp:9-29, sphere.hpp:10, box.hpp:9): kinds[i] 0 = sphere (cx cy cz r), 1 = box (min, max),
// 7 floats per child; names and material names joined by ';'
int ref_add_group(void* scene_ptr, const char* name, int n, const int* kinds, const float* params, const char* child_names, const char* material_names) {
	Scene* scene = static_cast<Scene*>(scene_ptr);
	auto split = [](std::string t) {
		std::vector<std::string> out;
		size_t cut;
		while ((cut = t.find(';')) != std::string::npos) {
			out.push_back(t.substr(0, cut));
			t = t.substr(cut + 1);
		}
		out.push_back(t);
		return out;
	};
	const std::vector<std::string> names = split(child_names), mats = split(material_names);
	if ((int) names.size() != n || (int) mats.size() != n) return 1;
	// "a/b/name": into Composite b inside Composite a under the root
	std::vector<std::shared_ptr<Composite>> above{scene->root};
	std::string own(name ? name : "");
	for (size_t cut; (cut = own.find('/')) != std::string::npos; own = own.substr(cut + 1)) {
		auto it = above.back()->children_.find(own.substr(0, cut));
		if (it == above.back()->children_.end()) return 1;
		auto next = std::dynamic_pointer_cast<Composite>(it->second);
		if (!next) return 1;
		above.push_back(next);
	}
	auto g = std::make_shared<Composite>(own);
	for (int i = 0; i < n; ++i) {
		const float* p = params + 7 * i;
		auto it = scene->materials.find(mats[(size_t) i]);
		if (it == scene->materials.end()) return 1;
		if (kinds[i] == 0) g->add_child(std::make_shared<Sphere>(p[3], glm::vec3{p[0], p[1], p[2]}, names[(size_t) i], it->second));
		else g->add_child(std::make_shared<Box>(glm::vec3{p[0], p[1], p[2]}, glm::vec3{p[3], p[4], p[5]}, names[(size_t) i], it->second));
	}
	g->build_octree();
	above.back()->add_child(g);
	for (size_t k = above.size(); k-- > 0;) above[k]->build_octree();
	return 0;
}
// scene.root->find_child(a)->find_child(b)...->translate / rotate / scale with the reference's own methods
// (composite.cpp:112-114, :168-181): path levels separated by '/', op 1 translate, 2 rotate (radians), 3 scale.
// (find_child itself dereferences end() for a name that is not there, so the name is looked up first)
int ref_transform_child(void* scene_ptr, const char* path, int op, float a, float b, float c) {
	Scene* scene = static_cast<Scene*>(scene_ptr);
	std::shared_ptr<Shape> shape = scene->root;
	std::string rest(path ? path : "");
	while (!rest.empty()) {
		const size_t cut = rest.find('/');
		const std::string name = rest.substr(0, cut);
		rest = cut == std::string::npos ? std::string() : rest.substr(cut + 1);
		auto comp = std::dynamic_pointer_cast<Composite>(shape);
		if (!comp || comp->children_.find(name) == comp->children_.end()) return 1;
		shape = comp->find_child(name);
	}
	if (op == 1) shape->translate(a, b, c);
	else if (op == 2) shape->rotate(a, b, c);
	else if (op == 3) shape->scale(a, b, c);
	else return 1;
	return 0;
}
void ref_flat_free(void* h) { delete static_cast<Flat*>(h); }
const char* ref_flat_error(void* h) { return static_cast<Flat*>(h)->error.c_str(); }
const void* ref_flat_array(void* h, int which, int* count, int* elem_bytes) {
	Flat* f = static_cast<Flat*>(h);
	switch (which) {
	case 0: *count = (int) f->frames.size(); *elem_bytes = (int) sizeof(PtxFrame); return f->frames.data();
	case 1: *count = (int) f->prims.size(); *elem_bytes = (int) sizeof(PtxPrim); return f->prims.data();
	case 2: *count = (int) f->materials.size(); *elem_bytes = (int) sizeof(PtxMaterial); return f->materials.data();
	case 3: *count = (int) f->lights.size(); *elem_bytes = (int) sizeof(PtxLight); return f->lights.data();
	case 4: *count = 3; *elem_bytes = 4; return f->ambient;
	case 5: *count = 1; *elem_bytes = (int) sizeof(PtxCamera); return &f->camera;
	case 6: *count = (int) f->guards.size(); *elem_bytes = (int) sizeof(PtxGuard); return f->guards.data();
	case 7: *count = (int) f->occurrences.size(); *elem_bytes = (int) sizeof(PtxOccurrence); return f->occurrences.data();
	}
	*count = 0;
	*elem_bytes = 0;
	return nullptr;
}
const char* ref_flat_name(void* h, int i) { return static_cast<Flat*>(h)->names[(size_t) i].c_str(); }

}  // extern "C"
