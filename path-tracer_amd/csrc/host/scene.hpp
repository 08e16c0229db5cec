// Host-side scene model with the reference's C++ API (same class and function names,
// same argument meaning), so that code written against the reference's headers
// (framework/scene.hpp, shape.hpp, sphere.hpp, box.hpp, triangle.hpp, composite.hpp,
// material.hpp, light.hpp, pointLight.hpp, camera.hpp, color.hpp, pixel.hpp) keeps
// compiling.  What is NOT here on purpose is Shape::intersect: intersection and shading
// run on the GPU only (include/ptx.h); the host tree exists to be loaded, transformed,
// octree-split exactly as the reference does it, and then flattened (flatten.hpp).
#pragma once
#include <iosfwd>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "linalg.hpp"

struct Color {  // framework/color.hpp:16-77
	float r = 0, g = 0, b = 0;
	Color& operator+=(Color const& o) { r += o.r; g += o.g; b += o.b; return *this; }
	Color& operator-=(Color const& o) { r -= o.r; g -= o.g; b -= o.b; return *this; }
	Color& operator*=(float f) { r *= f; g *= f; b *= f; return *this; }
	Color& operator*=(Color const& c) { r *= c.r; g *= c.g; b *= c.b; return *this; }
	friend Color operator+(Color a, Color const& b) { return a += b; }
	friend Color operator-(Color a, Color const& b) { return a -= b; }
	friend Color operator*(Color a, float f) { return a *= f; }
	friend Color operator*(Color a, Color const& c) { return a *= c; }
};

struct Material {  // framework/material.hpp:10-27
	std::string name = "default";
	Color ka{0.2f, 0.2f, 0.2f};
	Color kd{0.5f, 0.5f, 0.5f};
	Color ks{};
	float m = 0;
	float glossy = 0;
	float opacity = 1;
	float ior = 1;
};

struct Light {  // framework/light.hpp:7-12
	std::string name = "default";
	Color color{1, 1, 1};
	float brightness = 1;
	Color intensity = color * brightness;
};

struct PointLight {  // framework/pointLight.hpp:6-12
	std::string name = "default";
	Color color{255, 255, 255};
	vec3 position{};
	float brightness = 1;
	Color intensity = color * brightness;
};

struct Camera {  // framework/camera.hpp:8-14
	std::string name = "default";
	float fov_x = 60.0f;
	vec3 position{};
	vec3 direction{0, 0, -1};
	vec3 up{0, 1, 0};
};

struct Ray {  // framework/ray.hpp:7-14
	vec3 origin{0.0f, 0.0f, 0.0f};
	vec3 direction{0.0f, 0.0f, -1.0f};
};

struct HitPoint {  // framework/hitPoint.hpp:8-16
	bool does_intersect = false;
	float distance = 0.0f;
	std::string hit_object = "null";
	std::shared_ptr<Material> hit_material = nullptr;
	vec3 position{};
	vec3 ray_direction{};
	vec3 surface_normal{};
};

class Pixel {  // framework/pixel.hpp:19-29
public:
	Pixel(unsigned int a = 0, unsigned int b = 0) : x(a), y(b) {}
	unsigned int x = 0;
	unsigned int y = 0;
	Color color = {1.0, 1.0, 1.0};
};

class Shape {  // framework/shape.hpp:18-41
public:
	enum class Kind { sphere, box, triangle, composite };
	Shape(std::string const& name, std::shared_ptr<Material> material);
	virtual ~Shape() = default;
	virtual std::string get_name() const { return name_; }
	virtual float area() const = 0;
	virtual float volume() const = 0;
	virtual vec3 min(mat4 const& transform = mat4()) const = 0;
	virtual vec3 max(mat4 const& transform = mat4()) const = 0;
	virtual Kind kind() const = 0;

	virtual void transform(mat4 const& transformation);
	virtual void scale(float sx, float sy, float sz);
	virtual void rotate(float yaw, float pitch, float roll);
	virtual void translate(float x, float y, float z);

	mat4 const& world_transform() const { return world_transform_; }
	mat4 const& world_transform_inv() const { return world_transform_inv_; }
	std::shared_ptr<Material> const& material() const { return material_; }

protected:
	std::string name_;
	std::shared_ptr<Material> material_;
	mat4 world_transform_;
	mat4 world_transform_inv_;
};

class Sphere : public Shape {  // framework/sphere.hpp
public:
	Sphere(float radius = 1.0f, vec3 const& center = vec3(), std::string const& name = "sphere", std::shared_ptr<Material> material = nullptr);
	float area() const override;
	float volume() const override;
	vec3 min(mat4 const& transform = mat4()) const override;
	vec3 max(mat4 const& transform = mat4()) const override;
	Kind kind() const override { return Kind::sphere; }
	float radius() const { return radius_; }
	vec3 const& center() const { return center_; }

private:
	float radius_;
	vec3 center_;
};

class Box : public Shape {  // framework/box.hpp
public:
	Box(vec3 const& min = vec3(), vec3 const& max = vec3(), std::string const& name = "box", std::shared_ptr<Material> material = nullptr);
	float area() const override;
	float volume() const override;
	vec3 min(mat4 const& transform = mat4()) const override;
	vec3 max(mat4 const& transform = mat4()) const override;
	Kind kind() const override { return Kind::box; }
	float size_x() const { return max_.x - min_.x; }
	float size_y() const { return max_.y - min_.y; }
	float size_z() const { return max_.z - min_.z; }
	bool intersects_bounds(std::shared_ptr<Shape> const& shape) const;
	bool contains(vec3 const& v) const;
	vec3 const& min_corner() const { return min_; }
	vec3 const& max_corner() const { return max_; }

private:
	vec3 min_;
	vec3 max_;
};

class Triangle : public Shape {  // framework/triangle.hpp
public:
	Triangle(vec3 const& v0, vec3 const& v1, vec3 const& v2, std::string const& name = "triangle", std::shared_ptr<Material> material = nullptr);
	Triangle(vec3 const& v0, vec3 const& v1, vec3 const& v2, vec3 const& n, std::string const& name = "triangle",
	         std::shared_ptr<Material> material = nullptr);
	float area() const override;
	float volume() const override { return 0; }
	vec3 min(mat4 const& transform = mat4()) const override;
	vec3 max(mat4 const& transform = mat4()) const override;
	Kind kind() const override { return Kind::triangle; }
	vec3 const& v0() const { return v0_; }
	vec3 const& v1() const { return v1_; }
	vec3 const& v2() const { return v2_; }
	vec3 const& normal() const { return n_; }

private:
	vec3 v0_, v1_, v2_, n_;
	// min() / max() of the last (transform, world_transform_) pair asked for: an octree build asks
	// every triangle for both, under the same matrix, once per octant and level
	// (composite.cpp:141-147 through box.cpp:73-80), and would otherwise form the same matrix
	// product and the same three transformed corners over and over.  Same arithmetic, formed once.
	struct Bounds {
		mat4 transform, world;
		vec3 lo, hi;
		bool valid = false;
	};
	mutable Bounds bounds_;
	Bounds const& bounds(mat4 const& transform) const;
};

class Composite : public Shape {  // framework/composite.hpp
public:
	Composite(std::string const& name, std::shared_ptr<Material> material = nullptr);
	Composite(std::shared_ptr<Box> bounds, std::string const& name, std::shared_ptr<Material> material = nullptr);
	float area() const override;
	float volume() const override;
	vec3 min(mat4 const& transform = mat4()) const override;
	vec3 max(mat4 const& transform = mat4()) const override;
	Kind kind() const override { return Kind::composite; }

	void transform(mat4 const& transformation) override;
	void scale(float sx, float sy, float sz) override;
	void rotate(float yaw, float pitch, float roll) override;
	void translate(float x, float y, float z) override;

	void add_child(std::shared_ptr<Shape> shape);
	unsigned int child_count() { return (unsigned int) children_.size(); }
	std::shared_ptr<Shape> find_child(std::string const& name) const;
	void build_octree();

	std::shared_ptr<Box> const& bounds() const { return bounds_; }
	std::map<std::string, std::shared_ptr<Shape>> const& children() const { return children_; }

private:
	std::shared_ptr<Box> bounds_;
	std::map<std::string, std::shared_ptr<Shape>> children_;
};

struct Scene {  // framework/scene.hpp:13-21
	std::shared_ptr<Composite> root = std::make_shared<Composite>("root");
	std::map<std::string, std::shared_ptr<Material>> materials{};
	std::vector<PointLight> lights{};
	Light ambient{};
	Camera camera{};
	std::shared_ptr<Material> find_mat(std::string const& name) const;
};

// Thrown (as the reference throws `const char*`, framework/box.cpp:15) for input the
// reference answers with undefined behaviour: unknown material / shape names, face
// indices outside the vertex list, unreadable files.
struct SceneError {
	std::string message;
};

// framework/scene.hpp:24-35.  OBJ/MTL files are looked up in `obj_directory`; the
// one-argument form uses the folder of the scene file (the reference hard-codes
// "../../sdf/" relative to the working directory, framework/scene.cpp:326).
Scene load_scene(std::string const& file_path);
Scene load_scene(std::string const& file_path, std::string const& obj_directory);
std::map<std::string, std::shared_ptr<Material>> load_obj_materials(std::string const& file_path);
std::shared_ptr<Composite> load_obj(std::string const& directory_path, std::string const& name);
