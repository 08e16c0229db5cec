#include "flatten.hpp"

#include <cstring>
#include <unordered_map>

namespace {

bool is_identity(mat4 const& m) {
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r)
			if (m.at(c, r) != (c == r ? 1.0f : 0.0f)) return false;
	return true;
}

// true when everything below `c` is untransformed triangles in untransformed Composites:
// such a subtree can be handed to the GPU as one BVH
bool only_plain_triangles(Composite const& c) {
	for (auto const& kv : c.children()) {
		Shape const& s = *kv.second;
		if (!is_identity(s.world_transform())) return false;
		if (s.kind() == Shape::Kind::composite) {
			if (!only_plain_triangles(static_cast<Composite const&>(s))) return false;
		} else if (s.kind() != Shape::Kind::triangle) {
			return false;
		}
	}
	return !c.children().empty();
}

// true when `c` holds, at any depth below it but not below a transformed Composite, a leaf that is not an untransformed
// triangle (a sphere, a box, a triangle a caller moved).  Such a Composite cannot be melted into the mesh around it:
// Composite::intersect tests ITS bounds_ before it looks at the leaf (composite.cpp:77-82), and the leaf may have left
// them (nothing rebuilds them when a leaf moves), so it is handed over as a frame of its own, with its identity transform.
bool holds_other_leaves(Composite const& c) {
	for (auto const& kv : c.children()) {
		Shape const& s = *kv.second;
		if (s.kind() == Shape::Kind::composite) {
			if (is_identity(s.world_transform()) && holds_other_leaves(static_cast<Composite const&>(s))) return true;
		} else if (s.kind() != Shape::Kind::triangle || !is_identity(s.world_transform())) {
			return true;
		}
	}
	return false;
}

struct Walker {
	FlatScene& out;
	std::unordered_map<Shape const*, int> seen;
	std::unordered_map<Material const*, int> mat_index;

	int material_of(Shape const& s) {
		std::shared_ptr<Material> const& m = s.material();
		if (!m) throw SceneError{"shape '" + s.get_name() + "' has no material"};
		auto it = mat_index.find(m.get());
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
		mat_index.emplace(m.get(), idx);
		return idx;
	}

	void emit_leaf(Shape const& s, int frame, int levels_in, std::string const& path) {
		if (!seen.emplace(&s, (int) out.prims.size()).second) return;  // already listed at its first encounter
		PtxPrim p{};
		p.frame = frame;
		p.levels_in = levels_in;
		p.material = material_of(s);
		std::memcpy(p.world, s.world_transform().data(), sizeof p.world);
		std::memcpy(p.world_inv, s.world_transform_inv().data(), sizeof p.world_inv);
		switch (s.kind()) {
		case Shape::Kind::sphere: {
			auto const& q = static_cast<Sphere const&>(s);
			p.kind = PTX_SPHERE;
			p.p[0] = q.center().x; p.p[1] = q.center().y; p.p[2] = q.center().z; p.p[3] = q.radius();
			break;
		}
		case Shape::Kind::box: {
			auto const& q = static_cast<Box const&>(s);
			p.kind = PTX_BOX;
			p.p[0] = q.min_corner().x; p.p[1] = q.min_corner().y; p.p[2] = q.min_corner().z;
			p.p[3] = q.max_corner().x; p.p[4] = q.max_corner().y; p.p[5] = q.max_corner().z;
			break;
		}
		case Shape::Kind::triangle: {
			auto const& q = static_cast<Triangle const&>(s);
			p.kind = PTX_TRIANGLE;
			p.p[0] = q.v0().x; p.p[1] = q.v0().y; p.p[2] = q.v0().z;
			p.p[3] = q.v1().x; p.p[4] = q.v1().y; p.p[5] = q.v1().z;
			p.p[6] = q.v2().x; p.p[7] = q.v2().y; p.p[8] = q.v2().z;
			p.normal[0] = q.normal().x; p.normal[1] = q.normal().y; p.normal[2] = q.normal().z;
			break;
		}
		default: throw SceneError{"unexpected shape kind"};
		}
		out.prims.push_back(p);
		out.prim_names.push_back(s.get_name());
		out.prim_paths.push_back(path + s.get_name());
		out.prim_materials.push_back(s.material());
	}

	// the frame a transformed Composite becomes: `parent` the frame it lies in, `levels` the Composites between the two
	// (for a child of the root: up to the root, the root included)
	static PtxFrame frame_of(Composite const& sub, int parent, int levels) {
		if (!sub.bounds()) throw SceneError{"Composite '" + sub.get_name() + "' has no bounds: call build_octree()"};
		PtxFrame f{};
		std::memcpy(f.world, sub.world_transform().data(), sizeof f.world);
		std::memcpy(f.world_inv, sub.world_transform_inv().data(), sizeof f.world_inv);
		vec3 lo = sub.bounds()->min_corner(), hi = sub.bounds()->max_corner();
		f.bounds_min[0] = lo.x; f.bounds_min[1] = lo.y; f.bounds_min[2] = lo.z;
		f.bounds_max[0] = hi.x; f.bounds_max[1] = hi.y; f.bounds_max[2] = hi.z;
		f.parent = parent;
		f.levels_up = levels;
		return f;
	}

	// inside a mesh frame: `levels` Composites lie between the leaf and the frame's Composite
	void walk_mesh(Composite const& c, int frame, int levels, std::string const& path, int depth) {
		++out.composites;
		out.max_tree_depth = std::max(out.max_tree_depth, depth);
		for (auto const& kv : c.children()) {
			Shape const& s = *kv.second;
			if (s.kind() == Shape::Kind::composite) {
				auto const& sub = static_cast<Composite const&>(s);
				if (!is_identity(sub.world_transform()) || holds_other_leaves(sub)) {  // a transformed Composite inside a transformed one (or one that must keep its own bounds test): a frame inside this frame
					if (!seen.emplace(&s, -(int) out.frames.size()).second) continue;  // (listed at its first encounter, like a leaf)
					out.frames.push_back(frame_of(sub, frame, levels));
					walk_mesh(sub, (int) out.frames.size() - 1, 0, path + sub.get_name() + "/", depth + 1);
					continue;
				}
				walk_mesh(sub, frame, levels + 1, path + s.get_name() + "/", depth + 1);
			} else {
				++out.leaf_entries;
				// (a sphere, a box or a moved triangle in here is listed like any leaf; ptx_scene_create takes it when it
				// lies directly below the frame's Composite)
				emit_leaf(s, frame, levels, path);
			}
		}
	}

	// at root level: c is the root or one of its octants (`guard`: its index in out.guards, -1 for the root);
	// `levels` Composites from c up to the root, both included
	void walk_root(Composite const& c, int levels, std::string const& path, int depth, int guard) {
		++out.composites;
		out.max_tree_depth = std::max(out.max_tree_depth, depth);
		for (auto const& kv : c.children()) {
			Shape const& s = *kv.second;
			if (s.kind() != Shape::Kind::composite) {
				++out.leaf_entries;
				emit_leaf(s, 0, levels, path);
				// (a triangle that was first met inside an octant holding nothing but plain triangles belongs to that
				// octant's frame, which every ray that can hit it enters: its later entries need no listing)
				if (out.prims[(size_t) seen.at(&s)].frame == 0) out.occurrences.push_back(PtxOccurrence{guard, seen.at(&s), 0, 0});
				continue;
			}
			auto const& sub = static_cast<Composite const&>(s);
			if (is_identity(sub.world_transform()) && !only_plain_triangles(sub)) {  // an octant of the root
				if (!sub.bounds()) throw SceneError{"Composite '" + sub.get_name() + "' has no bounds: call build_octree()"};
				PtxGuard g{};
				vec3 lo = sub.bounds()->min_corner(), hi = sub.bounds()->max_corner();
				g.bounds_min[0] = lo.x; g.bounds_min[1] = lo.y; g.bounds_min[2] = lo.z;
				g.bounds_max[0] = hi.x; g.bounds_max[1] = hi.y; g.bounds_max[2] = hi.z;
				g.parent = guard;
				g.levels = levels + 1;
				int idx = (int) out.guards.size();
				out.guards.push_back(g);
				walk_root(sub, levels + 1, path + sub.get_name() + "/", depth + 1, idx);
				continue;
			}
			auto ins = seen.emplace(&s, -(int) out.frames.size());  // a mesh: remembered by (minus) its frame index
			if (!ins.second) {  // already turned into a frame from an earlier octant
				out.occurrences.push_back(PtxOccurrence{guard, -1, -ins.first->second, 0});
				continue;
			}
			int idx = (int) out.frames.size();
			out.frames.push_back(frame_of(sub, 0, levels));
			out.occurrences.push_back(PtxOccurrence{guard, -1, idx, 0});
			walk_mesh(sub, idx, 0, path + sub.get_name() + "/", depth + 1);
		}
	}
};

}  // namespace

FlatScene flatten_scene(Scene const& scene) {
	FlatScene out;
	if (!scene.root) throw SceneError{"scene has no root"};
	Composite const& root = *scene.root;
	if (!is_identity(root.world_transform())) throw SceneError{"the root Composite must not be transformed"};
	if (!root.bounds()) throw SceneError{"the root Composite has no bounds: call build_octree()"};
	PtxFrame f0{};
	std::memcpy(f0.world, root.world_transform().data(), sizeof f0.world);
	std::memcpy(f0.world_inv, root.world_transform_inv().data(), sizeof f0.world_inv);
	vec3 lo = root.bounds()->min_corner(), hi = root.bounds()->max_corner();
	f0.bounds_min[0] = lo.x; f0.bounds_min[1] = lo.y; f0.bounds_min[2] = lo.z;
	f0.bounds_max[0] = hi.x; f0.bounds_max[1] = hi.y; f0.bounds_max[2] = hi.z;
	f0.parent = -1;
	f0.levels_up = 0;
	out.frames.push_back(f0);

	Walker w{out, {}, {}};
	w.walk_root(root, 1, "", 0, -1);
	if (out.guards.empty()) out.occurrences.clear();  // a root that was not split: nothing to list

	for (PointLight const& L : scene.lights) {
		PtxLight pl{};
		pl.position[0] = L.position.x; pl.position[1] = L.position.y; pl.position[2] = L.position.z;
		pl.intensity[0] = L.intensity.r; pl.intensity[1] = L.intensity.g; pl.intensity[2] = L.intensity.b;
		out.lights.push_back(pl);
	}
	PtxSceneDesc& d = out.desc;
	d.frames = out.frames.data();
	d.n_frames = (int32_t) out.frames.size();
	d.prims = out.prims.data();
	d.n_prims = (int32_t) out.prims.size();
	d.materials = out.materials.data();
	d.n_materials = (int32_t) out.materials.size();
	d.lights = out.lights.data();
	d.n_lights = (int32_t) out.lights.size();
	d.guards = out.guards.empty() ? nullptr : out.guards.data();
	d.n_guards = (int32_t) out.guards.size();
	d.occurrences = out.occurrences.empty() ? nullptr : out.occurrences.data();
	d.n_occurrences = (int32_t) out.occurrences.size();
	d.ambient[0] = scene.ambient.intensity.r; d.ambient[1] = scene.ambient.intensity.g; d.ambient[2] = scene.ambient.intensity.b;
	d.camera.fov_x = scene.camera.fov_x;
	d.camera.position[0] = scene.camera.position.x; d.camera.position[1] = scene.camera.position.y; d.camera.position[2] = scene.camera.position.z;
	d.camera.direction[0] = scene.camera.direction.x; d.camera.direction[1] = scene.camera.direction.y; d.camera.direction[2] = scene.camera.direction.z;
	d.camera.up[0] = scene.camera.up.x; d.camera.up[1] = scene.camera.up.y; d.camera.up[2] = scene.camera.up.z;
	return out;
}
