// Host scene model: shapes, Composite/octree and the SDF / OBJ / MTL loaders, with the
// behaviour of the reference (framework/shape.cpp, sphere.cpp, box.cpp, triangle.cpp,
// composite.cpp, scene.cpp) including the quirks that decide which leaves exist and what
// bounds the GPU path filters rays with.
#include "scene.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

// ---------------------------------------------------------------- Shape (shape.cpp:3-36)
Shape::Shape(std::string const& name, std::shared_ptr<Material> material)
    : name_{name}, material_{std::move(material)}, world_transform_(), world_transform_inv_(inverse(world_transform_)) {}

// shape.cpp:13-15: sets the matrix and leaves the inverse stale, as the reference does
void Shape::transform(mat4 const& transformation) { world_transform_ = transformation; }
void Shape::scale(float sx, float sy, float sz) {
	world_transform_ = ::scale(world_transform_, vec3{sx, sy, sz});
	world_transform_inv_ = inverse(world_transform_);
}
void Shape::rotate(float yaw, float pitch, float roll) {
	world_transform_ = world_transform_ * euler_angle_yxz(yaw, pitch, roll);
	world_transform_inv_ = inverse(world_transform_);
}
void Shape::translate(float x, float y, float z) {
	world_transform_ = ::translate(world_transform_, vec3{x, y, z});
	world_transform_inv_ = inverse(world_transform_);
}

// ---------------------------------------------------------------- Sphere (sphere.cpp:9-28)
Sphere::Sphere(float radius, vec3 const& center, std::string const& name, std::shared_ptr<Material> material)
    : Shape(name, std::move(material)), radius_(std::fabs(radius)), center_(center) {}
float Sphere::area() const { return 4.0f * 3.14159265f * (float) std::pow((double) radius_, 2.0); }
float Sphere::volume() const { return 4.0f / 3.0f * 3.14159265f * std::fabs((float) std::pow((double) radius_, 3.0)); }
// the bounds ignore both the argument and the sphere's own transform (sphere.cpp:22-28)
vec3 Sphere::min(mat4 const&) const { return center_ - vec3{radius_, radius_, radius_}; }
vec3 Sphere::max(mat4 const&) const { return center_ + vec3{radius_, radius_, radius_}; }

// ---------------------------------------------------------------- Box (box.cpp:5-87)
Box::Box(vec3 const& min, vec3 const& max, std::string const& name, std::shared_ptr<Material> material)
    : Shape(name, std::move(material)), min_(min), max_(max) {
	if (min.x > max_.x || min.y > max_.y || min.z > max_.z) throw "Box minimum cannot be greater than maximum";
}
float Box::area() const { return 2 * size_x() * size_y() + 2 * size_y() * size_z() + 2 * size_z() * size_x(); }
float Box::volume() const { return size_x() * size_y() * size_z(); }
// box.cpp:42-53: the two extreme corners go through world_transform_ only
vec3 Box::min(mat4 const& transform) const {
	mat4 full = transform * world_transform_;
	vec3 r = min3(transform_vec(min_, world_transform_), transform_vec(max_, world_transform_));
	r = min3(r, transform_vec(vec3{min_.x, min_.y, max_.z}, full));
	r = min3(r, transform_vec(vec3{min_.x, max_.y, min_.z}, full));
	r = min3(r, transform_vec(vec3{min_.x, max_.y, max_.z}, full));
	r = min3(r, transform_vec(vec3{max_.x, min_.y, min_.z}, full));
	r = min3(r, transform_vec(vec3{max_.x, min_.y, max_.z}, full));
	r = min3(r, transform_vec(vec3{max_.x, max_.y, min_.z}, full));
	return r;
}
// box.cpp:55-66: seeded with the MINIMUM of the two extreme corners
vec3 Box::max(mat4 const& transform) const {
	mat4 full = transform * world_transform_;
	vec3 r = min3(transform_vec(min_, full), transform_vec(max_, full));
	r = max3(r, transform_vec(vec3{min_.x, min_.y, max_.z}, full));
	r = max3(r, transform_vec(vec3{min_.x, max_.y, min_.z}, full));
	r = max3(r, transform_vec(vec3{min_.x, max_.y, max_.z}, full));
	r = max3(r, transform_vec(vec3{max_.x, min_.y, min_.z}, full));
	r = max3(r, transform_vec(vec3{max_.x, min_.y, max_.z}, full));
	r = max3(r, transform_vec(vec3{max_.x, max_.y, min_.z}, full));
	return r;
}
bool Box::intersects_bounds(std::shared_ptr<Shape> const& shape) const {  // box.cpp:73-80
	vec3 lo = shape->min(world_transform_inv_);
	vec3 hi = shape->max(world_transform_inv_);
	auto axis = [](float a, float b, float mn, float mx) { return (a >= mn && a <= mx) || (b >= mn && b <= mx) || (a < mn && b > mx); };
	return axis(lo.x, hi.x, min_.x, max_.x) && axis(lo.y, hi.y, min_.y, max_.y) && axis(lo.z, hi.z, min_.z, max_.z);
}
bool Box::contains(vec3 const& v) const {
	return v.x >= min_.x && v.x <= max_.x && v.y >= min_.y && v.y <= max_.y && v.z >= min_.z && v.z <= max_.z;
}

// ---------------------------------------------------------------- Triangle (triangle.cpp:5-48)
Triangle::Triangle(vec3 const& v0, vec3 const& v1, vec3 const& v2, std::string const& name, std::shared_ptr<Material> material)
    : Triangle(v0, v1, v2, normalize(cross(v1 - v0, v2 - v0)), name, std::move(material)) {}
Triangle::Triangle(vec3 const& v0, vec3 const& v1, vec3 const& v2, vec3 const& n, std::string const& name, std::shared_ptr<Material> material)
    : Shape(name, std::move(material)), v0_{v0}, v1_{v1}, v2_{v2}, n_{n} {}
float Triangle::area() const {
	vec3 c = cross(v1_ - v0_, v2_ - v0_);
	return std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z) / 2;
}
Triangle::Bounds const& Triangle::bounds(mat4 const& transform) const {  // triangle.cpp:25-48
	Bounds& b = bounds_;
	if (b.valid && std::memcmp(&b.transform, &transform, sizeof(mat4)) == 0 && std::memcmp(&b.world, &world_transform_, sizeof(mat4)) == 0) return b;
	mat4 full = transform * world_transform_;
	vec3 a = transform_vec(v0_, full), c = transform_vec(v1_, full), e = transform_vec(v2_, full);
	b.lo = min3(min3(a, c), e);
	b.hi = max3(max3(a, c), e);
	b.transform = transform;
	b.world = world_transform_;
	b.valid = true;
	return b;
}
vec3 Triangle::min(mat4 const& transform) const { return bounds(transform).lo; }
vec3 Triangle::max(mat4 const& transform) const { return bounds(transform).hi; }

// ---------------------------------------------------------------- Composite (composite.cpp)
Composite::Composite(std::string const& name, std::shared_ptr<Material> material) : Shape(name, std::move(material)), bounds_{nullptr} {}
Composite::Composite(std::shared_ptr<Box> bounds, std::string const& name, std::shared_ptr<Material> material)
    : Shape(name, std::move(material)), bounds_{std::move(bounds)} {}
float Composite::area() const {
	float s = 0;
	for (auto const& kv : children_) s += kv.second->area();
	return s;
}
float Composite::volume() const {
	float s = 0;
	for (auto const& kv : children_) s += kv.second->volume();
	return s;
}
vec3 Composite::min(mat4 const& transform) const {  // composite.cpp:30-46
	if (bounds_) return bounds_->min(transform);
	vec3 r{};
	bool first = true;
	for (auto const& kv : children_) {
		vec3 c = kv.second->min(transform * world_transform_);
		r = first ? c : min3(r, c);
		first = false;
	}
	return r;
}
vec3 Composite::max(mat4 const& transform) const {  // composite.cpp:48-64
	if (bounds_) return bounds_->max(transform);
	vec3 r{};
	bool first = true;
	for (auto const& kv : children_) {
		vec3 c = kv.second->max(transform * world_transform_);
		r = first ? c : max3(r, c);
		first = false;
	}
	return r;
}
// composite.cpp:105-110: a second child of the same name is silently dropped
void Composite::add_child(std::shared_ptr<Shape> shape) { children_.emplace(shape->get_name(), shape); }
std::shared_ptr<Shape> Composite::find_child(std::string const& name) const {
	auto it = children_.find(name);
	if (it == children_.end()) throw SceneError{"no child named '" + name + "'"};  // composite.cpp:112-114 dereferences end()
	return it->second;
}
// composite.cpp:120-161: bounds from the children, then (above 64 children) eight equal
// octants filled by AABB overlap; the split is abandoned when an octant received every
// child, children that overlap no octant are lost, and every kept octant recurses.
void Composite::build_octree() {
	bounds_ = nullptr;
	vec3 lo = min(), hi = max();
	bounds_ = std::make_shared<Box>(lo, hi);
	if (children_.size() <= 64) return;
	vec3 base = min();
	vec3 half = (max() - base) * 0.5f;
	std::vector<std::shared_ptr<Composite>> octants;
	for (int x = 0; x < 2; ++x)
		for (int y = 0; y < 2; ++y)
			for (int z = 0; z < 2; ++z) {
				vec3 omin = base + vec3{x * half.x, y * half.y, z * half.z};
				vec3 omax = base + vec3{(x + 1) * half.x, (y + 1) * half.y, (z + 1) * half.z};
				octants.push_back(std::make_shared<Composite>(std::make_shared<Box>(omin, omax), std::to_string(z + 2 * y + 4 * x)));
			}
	for (auto const& oct : octants)
		for (auto const& kv : children_)
			if (oct->bounds_->intersects_bounds(kv.second)) oct->add_child(kv.second);
	for (auto const& oct : octants)
		if (oct->children_.size() == children_.size()) return;
	children_.clear();
	for (auto const& oct : octants)
		if (!oct->children_.empty()) {
			oct->build_octree();
			add_child(oct);
		}
}
void Composite::transform(mat4 const& t) { Shape::transform(t); build_octree(); }
void Composite::scale(float a, float b, float c) { Shape::scale(a, b, c); build_octree(); }
void Composite::rotate(float a, float b, float c) { Shape::rotate(a, b, c); build_octree(); }
void Composite::translate(float a, float b, float c) { Shape::translate(a, b, c); build_octree(); }

// ---------------------------------------------------------------- loading (scene.cpp)
std::shared_ptr<Material> Scene::find_mat(std::string const& name) const {
	auto it = materials.find(name);
	if (it == materials.end()) throw SceneError{"unknown material '" + name + "'"};  // scene.cpp:9-11 dereferences end()
	return it->second;
}

namespace {

// One text line split at white space.  Scalars whose token is missing are left
// uninitialised by the reference (e.g. `transform NAME scale 1.5`, scene.cpp:364-371); they
// read as NaN here, which makes the shape unhittable, the behaviour the reference build
// shows for such input.  Members of zero-initialised vectors / colours stay 0, members with
// a default keep it.
class Line {
public:
	explicit Line(std::string const& text) {
		std::istringstream s(text);
		std::string w;
		while (s >> w) words_.push_back(w);
	}
	std::string word() { return pos_ < words_.size() ? words_[pos_++] : std::string(); }
	float scalar() { return pos_ < words_.size() ? std::strtof(words_[pos_++].c_str(), nullptr) : std::numeric_limits<float>::quiet_NaN(); }
	void keep_default(float& f) { if (pos_ < words_.size()) f = std::strtof(words_[pos_++].c_str(), nullptr); }
	vec3 vec() { vec3 v; keep_default(v.x); keep_default(v.y); keep_default(v.z); return v; }
	Color color() { Color c; keep_default(c.r); keep_default(c.g); keep_default(c.b); return c; }

private:
	std::vector<std::string> words_;
	size_t pos_ = 0;
};

std::shared_ptr<Material> material_named(std::map<std::string, std::shared_ptr<Material>> const& mats, std::string const& name) {
	auto it = mats.find(name);
	if (it == mats.end()) throw SceneError{"unknown material '" + name + "'"};  // scene.cpp:52-53, :66-67, :80-81
	return it->second;
}

// scene.cpp:177-222: index groups "v", "v/vt", "v//vn", "v/vt/vn"; the face takes the
// normal of its FIRST vertex only
std::shared_ptr<Triangle> obj_face(Line& ln, std::vector<vec3> const& verts, std::vector<vec3> const& norms, std::string const& name,
                                   std::shared_ptr<Material> const& mat) {
	unsigned long iv[3] = {0, 0, 0}, in[3] = {0, 0, 0};
	bool has_normals = false;
	for (int i = 0; i < 3; ++i) {
		std::string group = ln.word();
		const char* s = group.c_str();
		char* end = nullptr;
		iv[i] = std::strtoul(s, &end, 10);
		if (*end != '/') continue;
		++end;
		if (*end != '/') std::strtoul(end, &end, 10);
		if (*end == '/') {
			in[i] = std::strtoul(end + 1, &end, 10);
			has_normals = true;
		}
	}
	for (int i = 0; i < 3; ++i)
		if (iv[i] < 1 || iv[i] > verts.size()) throw SceneError{"vertex index out of range in " + name};
	if (!has_normals) return std::make_shared<Triangle>(verts[iv[0] - 1], verts[iv[1] - 1], verts[iv[2] - 1], name, mat);
	if (in[0] < 1 || in[0] > norms.size()) throw SceneError{"normal index out of range in " + name};
	return std::make_shared<Triangle>(verts[iv[0] - 1], verts[iv[1] - 1], verts[iv[2] - 1], norms[in[0] - 1], name, mat);
}

std::string folder_of(std::string const& path) {
	auto slash = path.find_last_of('/');
	return slash == std::string::npos ? std::string("./") : path.substr(0, slash + 1);
}

}  // namespace

// scene.cpp:128-175
std::map<std::string, std::shared_ptr<Material>> load_obj_materials(std::string const& file_path) {
	std::map<std::string, std::shared_ptr<Material>> mats;
	std::ifstream in(file_path);
	std::string text;
	std::shared_ptr<Material> cur;
	while (std::getline(in, text)) {
		Line ln(text);
		std::string key = ln.word();
		if (key == "#") continue;
		if (key == "newmtl") {
			cur = std::make_shared<Material>();
			cur->name = ln.word();
			mats.emplace(cur->name, cur);
			continue;
		}
		if (!cur) continue;
		if (key == "Ka") cur->ka = ln.color();
		else if (key == "Kd") cur->kd = ln.color();
		else if (key == "Ks") cur->ks = ln.color();
		else if (key == "Ns") ln.keep_default(cur->m);
		else if (key == "d") ln.keep_default(cur->opacity);
		else if (key == "Ni") ln.keep_default(cur->ior);
		else if (key == "illum") {
			unsigned long model = std::strtoul(ln.word().c_str(), nullptr, 10);
			if (model == 2) cur->glossy = 1;          // highlight on: mirror
			else if (model == 7) cur->glossy = 0.01;  // refraction + Fresnel reflection
		}
	}
	return mats;
}

// scene.cpp:230-291
std::shared_ptr<Composite> load_obj(std::string const& directory_path, std::string const& name) {
	std::ifstream in(directory_path + name + ".obj");
	if (!in) throw SceneError{"cannot open " + directory_path + name + ".obj"};
	std::map<std::string, std::shared_ptr<Material>> mats;
	auto mesh = std::make_shared<Composite>(name, nullptr);
	std::shared_ptr<Composite> part = mesh;
	std::shared_ptr<Material> part_mat = std::make_shared<Material>();
	std::vector<vec3> verts, norms;
	int faces = 0;
	auto close_part = [&]() {
		if (mesh->get_name() != part->get_name()) {
			part->build_octree();
			mesh->add_child(part);
		}
	};
	std::string text;
	while (std::getline(in, text)) {
		Line ln(text);
		std::string key = ln.word();
		if (key == "#") continue;
		if (key == "mtllib") mats = load_obj_materials(directory_path + ln.word());
		else if (key == "o") {
			close_part();
			part = std::make_shared<Composite>(ln.word(), part_mat);
		} else if (key == "v") verts.push_back(ln.vec());
		else if (key == "vn") norms.push_back(ln.vec());
		else if (key == "usemtl") part_mat = material_named(mats, ln.word());
		else if (key == "f") {
			part->add_child(obj_face(ln, verts, norms, "face" + std::to_string(faces), part_mat));
			++faces;
		}
	}
	close_part();
	mesh->build_octree();
	return mesh;
}

Scene load_scene(std::string const& file_path) { return load_scene(file_path, folder_of(file_path)); }

// scene.cpp:375-400 with add_to_scene (:307-335) and transform (:337-373)
Scene load_scene(std::string const& file_path, std::string const& obj_directory) {
	Scene scene{};
	std::ifstream in(file_path);
	if (!in) throw SceneError{"cannot open " + file_path};
	std::string text;
	while (std::getline(in, text)) {
		Line ln(text);
		std::string key = ln.word();
		if (key == "#") continue;
		if (key == "define") {
			std::string what = ln.word();
			if (what == "material") {  // scene.cpp:25-41
				auto m = std::make_shared<Material>();
				m->name = ln.word();
				m->ka = ln.color();
				m->kd = ln.color();
				m->ks = ln.color();
				m->m = ln.scalar();
				m->glossy = ln.scalar();
				m->opacity = ln.scalar();
				m->ior = ln.scalar();
				scene.materials.emplace(m->name, m);
			} else if (what == "shape") {
				std::string kind = ln.word();
				if (kind == "box") {  // scene.cpp:43-54
					std::string name = ln.word();
					vec3 lo = ln.vec(), hi = ln.vec();
					try {
						scene.root->add_child(std::make_shared<Box>(lo, hi, name, material_named(scene.materials, ln.word())));
					} catch (const char* msg) {
						throw SceneError{std::string(msg) + " (" + name + ")"};
					}
				} else if (kind == "sphere") {  // scene.cpp:56-68
					std::string name = ln.word();
					vec3 c = ln.vec();
					float r = ln.scalar();
					scene.root->add_child(std::make_shared<Sphere>(r, c, name, material_named(scene.materials, ln.word())));
				} else if (kind == "triangle") {  // scene.cpp:70-82
					std::string name = ln.word();
					vec3 a = ln.vec(), b = ln.vec(), c = ln.vec();
					scene.root->add_child(std::make_shared<Triangle>(a, b, c, name, material_named(scene.materials, ln.word())));
				} else if (kind == "obj") {  // scene.cpp:323-326
					scene.root->add_child(load_obj(obj_directory, ln.word()));
				}
			} else if (what == "light") {  // scene.cpp:84-93
				std::string name = ln.word();
				vec3 pos = ln.vec();
				Color col = ln.color();
				float brightness = ln.scalar();
				scene.lights.push_back(PointLight{name, col, pos, brightness});
			} else if (what == "ambient") {  // scene.cpp:95-105: components are read in r, b, g order
				std::string name = ln.word();
				Color col;
				ln.keep_default(col.r);
				ln.keep_default(col.b);
				ln.keep_default(col.g);
				float brightness = ln.scalar();
				scene.ambient = Light{name, col, brightness};
			} else if (what == "camera") {  // scene.cpp:107-126
				Camera cam;
				cam.name = ln.word();
				cam.fov_x = ln.scalar();
				ln.keep_default(cam.position.x);
				ln.keep_default(cam.position.y);
				ln.keep_default(cam.position.z);
				float yaw = ln.scalar(), pitch = ln.scalar(), roll = ln.scalar();
				mat4 rot = euler_angle_yxz(radians(yaw), radians(pitch), radians(roll));
				cam.direction = transform_vec(vec3{0, 0, -1}, rot, false);
				cam.up = transform_vec(vec3{0, 1, 0}, rot, false);
				scene.camera = cam;
			}
		} else if (key == "transform") {
			std::string name = ln.word(), action = ln.word();
			std::shared_ptr<Shape> shape = scene.root->find_child(name);
			if (action == "translate") {
				float x = ln.scalar(), y = ln.scalar(), z = ln.scalar();
				shape->translate(x, y, z);
			} else if (action == "rotate") {
				float a = ln.scalar(), b = ln.scalar(), c = ln.scalar();
				shape->rotate(radians(a), radians(b), radians(c));
			} else if (action == "scale") {
				float a = ln.scalar(), b = ln.scalar(), c = ln.scalar();
				shape->scale(a, b, c);
			}
		}
		// "render <camera> <file> <w> <h>" is accepted and ignored (scene.cpp:293-305)
	}
	scene.root->build_octree();
	return scene;
}
