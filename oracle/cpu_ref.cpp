// ============================================================================
// oracle/cpu_ref.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference renderer's hot path (scene load -> octree ->
// Renderer::render -> Shape::intersect -> shading -> PPM), written from the
// reference's behaviour with its fp32 operation order kept, so that its output
// is bit-identical to the reference build (g++ -O2, x86-64, no FMA).
//
// Parity status: PINNED.  oracle/pin_oracle.py renders the golden scenes with
// this file and with the real reference (oracle/_ref/libref.so, built from the
// reference's own sources) and requires identical PPM bytes, identical float
// colour buffers and identical (hit, t, position, normal, ray_direction) for
// primary and random rays; tests/test_oracle.py re-checks this file against the
// committed golden vectors under tests/golden/.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file.  The product library never links it.
//
// Every function cites the reference file:line it restates (paths relative to
// the reference checkout).  Third-party arithmetic (GLM 0.9.5.3, vendored in the
// reference under external/glm-0.9.5.3/glm) is restated with its evaluation order.
// ============================================================================
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr float kEps = 0.001f;  // EPSILON, redefined per file: renderer.cpp:13 sphere.cpp:7 box.cpp:3 triangle.cpp:3

// ---------------------------------------------------------------------------
// GLM subset (evaluation order matters: these are NOT "any correct formula")
// ---------------------------------------------------------------------------
struct V3 {
	float x = 0, y = 0, z = 0;  // glm::tvec3() zero-initialises (detail/type_vec3.inl:59-63)
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
// detail/func_geometric.inl:61-68  (tmp.x + tmp.y + tmp.z, left to right)
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// detail/func_geometric.inl:212-224
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
// detail/func_geometric.inl:104-110
inline float length(V3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
// detail/func_geometric.inl:252-261 with inversesqrt = 1.0f / sqrt(x) (func_exponential.inl:216-219)
inline V3 normalize(V3 v) {
	float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
	return v * (1.0f / std::sqrt(sqr));
}
// detail/func_common.inl:365-400: min = x < y ? x : y, max = x < y ? y : x
inline float gmin(float x, float y) { return x < y ? x : y; }
inline float gmax(float x, float y) { return x < y ? y : x; }
inline V3 vmin(V3 a, V3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

struct V4 {
	float x, y, z, w;
};
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

struct M4 {
	V4 c[4];  // columns, as glm::mat4
};
inline M4 identity() {  // detail/type_mat4x4.inl:67-75
	return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
// mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*w)   detail/type_mat4x4.inl:676-687
inline V4 mul(M4 const& m, V4 v) {
	V4 add0 = m.c[0] * v.x + m.c[1] * v.y;
	V4 add1 = m.c[2] * v.z + m.c[3] * v.w;
	return add0 + add1;
}
// mat4 * mat4: ((a0*b0 + a1*b1) + a2*b2) + a3*b3 per column   detail/type_mat4x4.inl:753-775
inline M4 mul(M4 const& a, M4 const& b) {
	M4 r;
	for (int j = 0; j < 4; ++j) {
		V4 bj = b.c[j];
		r.c[j] = a.c[0] * bj.x + a.c[1] * bj.y + a.c[2] * bj.z + a.c[3] * bj.w;
	}
	return r;
}
// detail/type_mat4x4.inl:476-535
inline M4 inverse(M4 const& m) {
	auto e = [&](int col, int row) -> float {
		V4 const& c = m.c[col];
		return row == 0 ? c.x : row == 1 ? c.y : row == 2 ? c.z : c.w;
	};
	float Coef00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3);
	float Coef02 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3);
	float Coef03 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
	float Coef04 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3);
	float Coef06 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3);
	float Coef07 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
	float Coef08 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2);
	float Coef10 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2);
	float Coef11 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
	float Coef12 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3);
	float Coef14 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3);
	float Coef15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
	float Coef16 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2);
	float Coef18 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2);
	float Coef19 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
	float Coef20 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1);
	float Coef22 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1);
	float Coef23 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);
	V4 Fac0{Coef00, Coef00, Coef02, Coef03};
	V4 Fac1{Coef04, Coef04, Coef06, Coef07};
	V4 Fac2{Coef08, Coef08, Coef10, Coef11};
	V4 Fac3{Coef12, Coef12, Coef14, Coef15};
	V4 Fac4{Coef16, Coef16, Coef18, Coef19};
	V4 Fac5{Coef20, Coef20, Coef22, Coef23};
	V4 Vec0{e(1, 0), e(0, 0), e(0, 0), e(0, 0)};
	V4 Vec1{e(1, 1), e(0, 1), e(0, 1), e(0, 1)};
	V4 Vec2{e(1, 2), e(0, 2), e(0, 2), e(0, 2)};
	V4 Vec3{e(1, 3), e(0, 3), e(0, 3), e(0, 3)};
	V4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
	V4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
	V4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
	V4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;
	V4 SignA{+1, -1, +1, -1};
	V4 SignB{-1, +1, -1, +1};
	M4 inv{{Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB}};
	V4 Row0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
	V4 Dot0 = m.c[0] * Row0;
	float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
	float OneOverDeterminant = 1.0f / Dot1;
	return {{inv.c[0] * OneOverDeterminant, inv.c[1] * OneOverDeterminant, inv.c[2] * OneOverDeterminant,
	         inv.c[3] * OneOverDeterminant}};
}
// gtc/matrix_transform.inl:36-45
inline M4 translate(M4 const& m, V3 v) {
	M4 r = m;
	r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
	return r;
}
// gtc/matrix_transform.inl:149-161
inline M4 scale(M4 const& m, V3 v) { return {{m.c[0] * v.x, m.c[1] * v.y, m.c[2] * v.z, m.c[3]}}; }
// gtx/euler_angles.inl:119-151
inline M4 euler_yxz(float yaw, float pitch, float roll) {
	float ch = std::cos(yaw), sh = std::sin(yaw);
	float cp = std::cos(pitch), sp = std::sin(pitch);
	float cb = std::cos(roll), sb = std::sin(roll);
	M4 r;
	r.c[0] = {ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0};
	r.c[1] = {-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0};
	r.c[2] = {sh * cp, -sp, ch * cp, 0};
	r.c[3] = {0, 0, 0, 1};
	return r;
}
// detail/func_trigonometric.inl:35-45
inline float radians(float deg) { return deg * float(0.01745329251994329576923690768489); }

// framework/shape.cpp:46-48  transform_vec
inline V3 xform(V3 v, M4 const& m, bool is_location = true) {
	V4 r = mul(m, V4{v.x, v.y, v.z, is_location ? 1.0f : 0.0f});
	return {r.x, r.y, r.z};
}

// ---------------------------------------------------------------------------
// value types (framework/color.hpp, material.hpp, ray.hpp, hitPoint.hpp, light.hpp,
// pointLight.hpp, camera.hpp)
// ---------------------------------------------------------------------------
struct Color {
	float r = 0, g = 0, b = 0;
	Color& operator+=(Color o) { r += o.r; g += o.g; b += o.b; return *this; }
	Color& operator*=(float f) { r *= f; g *= f; b *= f; return *this; }
	Color& operator*=(Color o) { r *= o.r; g *= o.g; b *= o.b; return *this; }
};
inline Color operator*(Color a, float f) { a *= f; return a; }
inline Color operator*(Color a, Color b) { a *= b; return a; }

struct Material {  // material.hpp:10-27
	std::string name = "default";
	Color ka{0.2f, 0.2f, 0.2f};
	Color kd{0.5f, 0.5f, 0.5f};
	Color ks{};
	float m = 0, glossy = 0, opacity = 1, ior = 1;
};

struct Ray {  // ray.hpp:7-14
	V3 o{0, 0, 0}, d{0, 0, -1};
	V3 point(float t) const { return o + d * t; }
};
// framework/shape.cpp:50-52
inline Ray xform_ray(Ray const& r, M4 const& m) { return {xform(r.o, m), xform(r.d, m, false)}; }

struct Shape;
struct Hit {  // hitPoint.hpp:8-16 (name and material are reached through `leaf`)
	bool does = false;
	float distance = 0;
	Shape const* leaf = nullptr;
	V3 position, ray_direction, normal;
};

struct PointLight {  // pointLight.hpp:6-12; intensity = color * brightness
	V3 position;
	Color intensity;
};
struct Camera {  // camera.hpp:8-14
	float fov_x = 60.0f;
	V3 position{}, direction{0, 0, -1}, up{0, 1, 0};
};

struct Counters {
	uint64_t trace_rays = 0, shadow_rays = 0, shaded_hits = 0;
	uint64_t sphere_tests = 0, box_tests = 0, tri_tests = 0, bounds_tests = 0;
};
thread_local Counters* tl_counters = nullptr;

// ---------------------------------------------------------------------------
// Shape hierarchy (framework/shape.hpp:18-41)
// ---------------------------------------------------------------------------
struct Shape {
	std::string name;
	std::shared_ptr<Material> material;
	M4 W = identity();
	M4 Winv = inverse(identity());  // shape.cpp:3-7: the inverse is COMPUTED, so it may carry -0 entries
	Shape(std::string n, std::shared_ptr<Material> m) : name(std::move(n)), material(std::move(m)) {}
	virtual ~Shape() = default;
	virtual V3 bmin(M4 const& t) const = 0;
	virtual V3 bmax(M4 const& t) const = 0;
	virtual Hit intersect(Ray const& ray) const = 0;
	// shape.cpp:17-36: post-multiply, then recompute the inverse
	virtual void do_scale(float sx, float sy, float sz) { W = scale(W, {sx, sy, sz}); Winv = inverse(W); }
	virtual void do_rotate(float yaw, float pitch, float roll) { W = mul(W, euler_yxz(yaw, pitch, roll)); Winv = inverse(W); }
	virtual void do_translate(float x, float y, float z) { W = translate(W, {x, y, z}); Winv = inverse(W); }
};

struct Sphere : Shape {  // framework/sphere.cpp
	float radius;
	V3 center;
	Sphere(float r, V3 c, std::string n, std::shared_ptr<Material> m) : Shape(std::move(n), std::move(m)), radius(std::fabs(r)), center(c) {}
	// sphere.cpp:22-28: the transform argument AND world_transform_ are ignored
	V3 bmin(M4 const&) const override { return center - V3{radius, radius, radius}; }
	V3 bmax(M4 const&) const override { return center + V3{radius, radius, radius}; }
	// sphere.cpp:35-52 + glm gtx/intersect.inl:152-170
	Hit intersect(Ray const& ray) const override {
		if (tl_counters) ++tl_counters->sphere_tests;
		Ray ri = xform_ray(ray, Winv);
		V3 dn = normalize(ri.d);
		float r2 = radius * radius;
		const float feps = std::numeric_limits<float>::epsilon();
		V3 diff = center - ri.o;
		float t0 = dot(diff, dn);
		float dsq = dot(diff, diff) - t0 * t0;
		if (dsq > r2) return {};
		float t1 = std::sqrt(r2 - dsq);
		float t = t0 > t1 + feps ? t0 - t1 : t0 + t1;
		if (!(t > feps)) return {};
		t /= length(ri.d);
		t -= kEps;
		V3 pos = xform(ri.point(t), W);
		V3 n = normalize(pos - xform(center, W));  // sphere.cpp:54-56
		return {true, t, this, pos, ray.d, n};
	}
};

struct Box : Shape {  // framework/box.cpp
	V3 lo, hi;
	Box(V3 mn, V3 mx, std::string n = "box", std::shared_ptr<Material> m = nullptr) : Shape(std::move(n), std::move(m)), lo(mn), hi(mx) {
		if (mn.x > mx.x || mn.y > mx.y || mn.z > mx.z) throw "Box minimum cannot be greater than maximum";  // box.cpp:14-16
	}
	// box.cpp:42-53: the first two corners go through world_transform_ only (not `transform`)
	V3 bmin(M4 const& t) const override {
		M4 f = mul(t, W);
		V3 m = vmin(xform(lo, W), xform(hi, W));
		m = vmin(m, xform({lo.x, lo.y, hi.z}, f));
		m = vmin(m, xform({lo.x, hi.y, lo.z}, f));
		m = vmin(m, xform({lo.x, hi.y, hi.z}, f));
		m = vmin(m, xform({hi.x, lo.y, lo.z}, f));
		m = vmin(m, xform({hi.x, lo.y, hi.z}, f));
		m = vmin(m, xform({hi.x, hi.y, lo.z}, f));
		return m;
	}
	// box.cpp:55-66: starts from glm::MIN of the two extreme corners, then max over six others
	V3 bmax(M4 const& t) const override {
		M4 f = mul(t, W);
		V3 m = vmin(xform(lo, f), xform(hi, f));
		m = vmax(m, xform({lo.x, lo.y, hi.z}, f));
		m = vmax(m, xform({lo.x, hi.y, lo.z}, f));
		m = vmax(m, xform({lo.x, hi.y, hi.z}, f));
		m = vmax(m, xform({hi.x, lo.y, lo.z}, f));
		m = vmax(m, xform({hi.x, lo.y, hi.z}, f));
		m = vmax(m, xform({hi.x, hi.y, lo.z}, f));
		return m;
	}
	// box.cpp:73-80
	bool intersects_bounds(Shape const& s) const {
		V3 mn = s.bmin(Winv), mx = s.bmax(Winv);
		return ((mn.x >= lo.x && mn.x <= hi.x) || (mx.x >= lo.x && mx.x <= hi.x) || (mn.x < lo.x && mx.x > hi.x)) &&
		       ((mn.y >= lo.y && mn.y <= hi.y) || (mx.y >= lo.y && mx.y <= hi.y) || (mn.y < lo.y && mx.y > hi.y)) &&
		       ((mn.z >= lo.z && mn.z <= hi.z) || (mx.z >= lo.z && mx.z <= hi.z) || (mn.z < lo.z && mx.z > hi.z));
	}
	// box.cpp:105-165  slab test on an already-local ray; also every Composite's bounds test
	bool slab(Ray const& r, float& t) const {
		float t_min = -std::numeric_limits<float>::infinity();
		float t_max = std::numeric_limits<float>::infinity();
		if (0 == r.d.x) {
			if (r.o.x < lo.x || r.o.x > hi.x) return false;
		} else {
			float inv = 1 / r.d.x;
			float t1 = (lo.x - r.o.x) * inv, t2 = (hi.x - r.o.x) * inv;
			t_min = std::max(t_min, std::min(t1, t2));
			t_max = std::min(t_max, std::max(t1, t2));
		}
		if (0 == r.d.y) {
			if (r.o.y < lo.y || r.o.y > hi.y) return false;
		} else {
			float inv = 1 / r.d.y;
			float t1 = (lo.y - r.o.y) * inv, t2 = (hi.y - r.o.y) * inv;
			t_min = std::max(t_min, std::min(t1, t2));
			t_max = std::min(t_max, std::max(t1, t2));
		}
		if (0 == r.d.z) {
			if (r.o.z < lo.z || r.o.z > hi.z) return false;
		} else {
			float inv = 1 / r.d.z;
			float t1 = (lo.z - r.o.z) * inv, t2 = (hi.z - r.o.z) * inv;
			t_min = std::max(t_min, std::min(t1, t2));
			t_max = std::min(t_max, std::max(t1, t2));
		}
		if (t_max < t_min) return false;
		if (t_min > 0) t = t_min;
		else if (t_max > 0) t = t_max;
		else return false;
		t -= kEps;
		return true;
	}
	// box.cpp:167-182: first match wins, tolerance 2*EPSILON, falls through to the zero vector
	V3 face_normal(V3 p) const {
		if (p.x <= lo.x + kEps * 2) return {-1, 0, 0};
		else if (p.y <= lo.y + kEps * 2) return {0, -1, 0};
		else if (p.z <= lo.z + kEps * 2) return {0, 0, -1};
		else if (p.x >= hi.x - kEps * 2) return {1, 0, 0};
		else if (p.y >= hi.y - kEps * 2) return {0, 1, 0};
		else if (p.z >= hi.z - kEps * 2) return {0, 0, 1};
		return {};
	}
	// box.cpp:89-102
	Hit intersect(Ray const& ray) const override {
		if (tl_counters) ++tl_counters->box_tests;
		float t;
		Ray ri = xform_ray(ray, Winv);
		if (!slab(ri, t)) return {};
		V3 nl = face_normal(ri.point(t));
		V3 nw = xform(nl, W, false);
		return {true, t, this, ray.point(t), ray.d, normalize(nw)};
	}
};

struct Triangle : Shape {  // framework/triangle.cpp
	V3 v0, v1, v2, n;
	Triangle(V3 a, V3 b, V3 c, V3 nn, std::string nm, std::shared_ptr<Material> m) : Shape(std::move(nm), std::move(m)), v0(a), v1(b), v2(c), n(nn) {}
	// triangle.cpp:5-11: geometric normal when none is given
	Triangle(V3 a, V3 b, V3 c, std::string nm, std::shared_ptr<Material> m) : Triangle(a, b, c, normalize(cross(b - a, c - a)), std::move(nm), std::move(m)) {}
	V3 bmin(M4 const& t) const override {  // triangle.cpp:36-41
		M4 f = mul(t, W);
		V3 m = vmin(xform(v0, f), xform(v1, f));
		return vmin(m, xform(v2, f));
	}
	V3 bmax(M4 const& t) const override {  // triangle.cpp:43-48
		M4 f = mul(t, W);
		V3 m = vmax(xform(v0, f), xform(v1, f));
		return vmax(m, xform(v2, f));
	}
	// triangle.cpp:56-88  Moeller-Trumbore, two-sided, inv_det multiply
	Hit intersect(Ray const& ray) const override {
		if (tl_counters) ++tl_counters->tri_tests;
		Ray ri = xform_ray(ray, Winv);
		V3 e1 = v1 - v0, e2 = v2 - v0;
		V3 p = cross(ri.d, e2);
		float det = dot(e1, p);
		if (det < kEps && det > -kEps) return {};
		float inv_det = 1 / det;
		V3 tv = ri.o - v0;
		float u = dot(tv, p) * inv_det;
		if (u < 0 || u > 1) return {};
		V3 q = cross(tv, e1);
		float v = dot(ri.d, q) * inv_det;
		if (v < 0 || u + v > 1) return {};
		float t = dot(e2, q) * inv_det;
		if (t < kEps) return {};
		t -= kEps;
		// position on the incoming ray, LOCAL direction, normal neither flipped nor normalised
		return {true, t, this, ray.point(t), ri.d, xform(n, W, false)};
	}
};

struct Composite : Shape {  // framework/composite.cpp
	std::shared_ptr<Box> bounds;
	std::map<std::string, std::shared_ptr<Shape>> children;  // composite.hpp:35: name-ordered
	Composite(std::string n, std::shared_ptr<Material> m = nullptr) : Shape(std::move(n), std::move(m)) {}
	Composite(std::shared_ptr<Box> b, std::string n) : Shape(std::move(n), nullptr), bounds(std::move(b)) {}
	// composite.cpp:30-64
	V3 bmin(M4 const& t) const override {
		if (bounds) return bounds->bmin(t);
		V3 m{};
		bool first = true;
		for (auto const& it : children) {
			V3 c = it.second->bmin(mul(t, W));
			m = first ? c : vmin(m, c);
			first = false;
		}
		return m;
	}
	V3 bmax(M4 const& t) const override {
		if (bounds) return bounds->bmax(t);
		V3 m{};
		bool first = true;
		for (auto const& it : children) {
			V3 c = it.second->bmax(mul(t, W));
			m = first ? c : vmax(m, c);
			first = false;
		}
		return m;
	}
	void add_child(std::shared_ptr<Shape> s) { children.emplace(s->name, s); }  // composite.cpp:105-110: duplicates dropped
	// composite.cpp:75-103
	Hit intersect(Ray const& ray) const override {
		if (tl_counters) ++tl_counters->bounds_tests;
		float t;
		if (!bounds || !bounds->slab(ray, t)) return {};
		Ray ri = xform_ray(ray, Winv);
		float min_t = -1;
		Hit best{};
		for (auto const& it : children) {
			Hit h = it.second->intersect(ri);
			if (!h.does) continue;
			if (!best.does || h.distance < min_t) {
				min_t = h.distance;
				best = h;
			}
		}
		if (best.does) {
			best.position = xform(best.position, W);
			best.normal = normalize(xform(best.normal, W, false));
		}
		return best;
	}
	// composite.cpp:120-161
	void build_octree() {
		bounds = nullptr;
		V3 mn0 = bmin(identity()), mx0 = bmax(identity());
		bounds = std::make_shared<Box>(mn0, mx0);
		if (children.size() <= 64) return;
		V3 mn = bmin(identity()), mx = bmax(identity());  // now answered by the bounds box
		V3 sz = (mx - mn) * 0.5f;
		std::vector<std::shared_ptr<Composite>> subs;
		for (int x = 0; x < 2; ++x)
			for (int y = 0; y < 2; ++y)
				for (int z = 0; z < 2; ++z) {
					V3 omin = mn + V3{x * sz.x, y * sz.y, z * sz.z};
					V3 omax = mn + V3{(x + 1) * sz.x, (y + 1) * sz.y, (z + 1) * sz.z};
					subs.push_back(std::make_shared<Composite>(std::make_shared<Box>(omin, omax), std::to_string(z + 2 * y + 4 * x)));
				}
		for (auto const& oct : subs)
			for (auto const& it : children)
				if (oct->bounds->intersects_bounds(*it.second)) oct->add_child(it.second);
		for (auto const& oct : subs)
			if (oct->children.size() == children.size()) return;
		children.clear();
		for (auto const& oct : subs)
			if (!oct->children.empty()) {
				oct->build_octree();
				add_child(oct);
			}
	}
	// composite.cpp:168-181
	void do_scale(float a, float b, float c) override { Shape::do_scale(a, b, c); build_octree(); }
	void do_rotate(float a, float b, float c) override { Shape::do_rotate(a, b, c); build_octree(); }
	void do_translate(float a, float b, float c) override { Shape::do_translate(a, b, c); build_octree(); }
};

struct Scene {  // framework/scene.hpp:13-21
	std::shared_ptr<Composite> root = std::make_shared<Composite>("root");
	std::map<std::string, std::shared_ptr<Material>> materials;
	std::vector<PointLight> lights;
	Color ambient{1, 1, 1};  // light.hpp:7-12 default intensity = (1,1,1)*1
	Camera camera;
	std::string error;
	// rank bookkeeping for the tests (first depth-first encounter order of every leaf)
	std::vector<Shape const*> leaves;
	std::map<Shape const*, int> rank;
	std::vector<std::string> leaf_path;
};

// ---------------------------------------------------------------------------
// Scene loading (framework/scene.cpp).  Tokens are whitespace separated; a float whose
// token is missing is left uninitialised by the reference (scene.cpp:364-371 with
// example.sdf:25); here it is NaN, which reproduces what the reference build shows for
// that file (inverse full of NaN => shape never hit).  glm::vec3 / Color members that
// are missing stay 0 (they are zero-initialised before extraction).
// ---------------------------------------------------------------------------
struct Toks {
	std::vector<std::string> t;
	size_t i = 0;
	explicit Toks(std::string const& line) {
		std::istringstream s(line);
		std::string w;
		while (s >> w) t.push_back(w);
	}
	bool more() const { return i < t.size(); }
	std::string word() { return more() ? t[i++] : std::string(); }
	float scalar() { return more() ? std::strtof(t[i++].c_str(), nullptr) : std::numeric_limits<float>::quiet_NaN(); }
	float member() { return more() ? std::strtof(t[i++].c_str(), nullptr) : 0.0f; }
	void into(float& f) { if (more()) f = std::strtof(t[i++].c_str(), nullptr); }  // target keeps its default
	V3 vec() { V3 v; v.x = member(); v.y = member(); v.z = member(); return v; }
	Color color() { Color c; c.r = member(); c.g = member(); c.b = member(); return c; }
};

std::shared_ptr<Material> find_material(Scene& sc, std::string const& n) {
	auto it = sc.materials.find(n);
	if (it == sc.materials.end()) {  // scene.cpp:52-53: the reference dereferences end() here (UB)
		sc.error = "unknown material '" + n + "'";
		return std::make_shared<Material>();
	}
	return it->second;
}

// scene.cpp:128-175
std::map<std::string, std::shared_ptr<Material>> load_obj_materials(std::string const& path) {
	std::map<std::string, std::shared_ptr<Material>> mats;
	std::ifstream in(path);
	std::string line;
	std::shared_ptr<Material> cur;
	while (std::getline(in, line)) {
		Toks k(line);
		std::string tok = k.word();
		if (tok == "#") continue;
		if (tok == "newmtl") {
			cur = std::make_shared<Material>();
			cur->name = k.word();
			mats.emplace(cur->name, cur);
		} else if (!cur) {
			continue;
		} else if (tok == "Ka") cur->ka = k.color();
		else if (tok == "Kd") cur->kd = k.color();
		else if (tok == "Ks") cur->ks = k.color();
		else if (tok == "Ns") k.into(cur->m);
		else if (tok == "illum") {
			unsigned model = (unsigned) std::strtoul(k.word().c_str(), nullptr, 10);
			if (model == 2) cur->glossy = 1;
			else if (model == 7) cur->glossy = 0.01;
		} else if (tok == "d") k.into(cur->opacity);
		else if (tok == "Ni") k.into(cur->ior);
	}
	return mats;
}

// scene.cpp:177-222: "v", "v/vt", "v//vn", "v/vt/vn"; only the FIRST vertex's normal is used
std::shared_ptr<Triangle> load_obj_face(Toks& k, std::vector<V3> const& verts, std::vector<V3> const& norms,
                                        std::string const& name, std::shared_ptr<Material> mat, std::string& err) {
	unsigned iv[3] = {0, 0, 0}, in[3] = {0, 0, 0};
	bool has_n = false;
	for (int i = 0; i < 3; ++i) {
		std::string g = k.word();
		const char* s = g.c_str();
		char* e;
		iv[i] = (unsigned) std::strtoul(s, &e, 10);
		if (*e == '/') {
			++e;
			if (*e != '/') std::strtoul(e, &e, 10);  // texture index, unused
			if (*e == '/') {
				++e;
				in[i] = (unsigned) std::strtoul(e, &e, 10);
				has_n = true;
			}
		}
	}
	for (int i = 0; i < 3; ++i)
		if (iv[i] < 1 || iv[i] > verts.size()) { err = "face index out of range in " + name; return nullptr; }
	if (has_n) {
		if (in[0] < 1 || in[0] > norms.size()) { err = "normal index out of range in " + name; return nullptr; }
		return std::make_shared<Triangle>(verts[iv[0] - 1], verts[iv[1] - 1], verts[iv[2] - 1], norms[in[0] - 1], name, mat);
	}
	return std::make_shared<Triangle>(verts[iv[0] - 1], verts[iv[1] - 1], verts[iv[2] - 1], name, mat);
}

// scene.cpp:230-291
std::shared_ptr<Composite> load_obj(std::string const& dir, std::string const& name, std::string& err) {
	std::ifstream in(dir + name + ".obj");
	if (!in) { err = "cannot open " + dir + name + ".obj"; }
	std::map<std::string, std::shared_ptr<Material>> mats;
	auto comp = std::make_shared<Composite>(name, nullptr);
	std::shared_ptr<Composite> cur = comp;
	std::shared_ptr<Material> child_mat = std::make_shared<Material>();
	std::vector<V3> verts, norms;
	int face_count = 0;
	std::string line;
	while (std::getline(in, line)) {
		Toks k(line);
		std::string tok = k.word();
		if (tok == "#") continue;
		if (tok == "mtllib") {
			mats = load_obj_materials(dir + k.word());
		} else if (tok == "o") {
			if (comp->name != cur->name) {
				cur->build_octree();
				comp->add_child(cur);
			}
			cur = std::make_shared<Composite>(k.word(), child_mat);
		} else if (tok == "v") verts.push_back(k.vec());
		else if (tok == "vn") norms.push_back(k.vec());
		else if (tok == "usemtl") {
			auto it = mats.find(k.word());
			if (it == mats.end()) { err = "unknown OBJ material in " + name; }
			else child_mat = it->second;
		} else if (tok == "f") {
			auto tri = load_obj_face(k, verts, norms, "face" + std::to_string(face_count), child_mat, err);
			if (tri) cur->add_child(tri);
			++face_count;
		}
	}
	if (comp->name != cur->name) {
		cur->build_octree();
		comp->add_child(cur);
	}
	comp->build_octree();
	return comp;
}

// scene.cpp:375-400 (+ :307-373).  obj_dir replaces the literal "../../sdf/" of scene.cpp:326.
std::unique_ptr<Scene> load_scene(std::string const& path, std::string const& obj_dir) {
	auto sc = std::make_unique<Scene>();
	std::ifstream in(path);
	if (!in) { sc->error = "cannot open " + path; return sc; }
	std::string line;
	while (std::getline(in, line)) {
		Toks k(line);
		std::string tok = k.word();
		if (tok == "#") continue;
		if (tok == "define") {
			std::string what = k.word();
			if (what == "material") {  // scene.cpp:25-41
				auto m = std::make_shared<Material>();
				m->name = k.word();
				m->ka = k.color(); m->kd = k.color(); m->ks = k.color();
				m->m = k.scalar(); m->glossy = k.scalar(); m->opacity = k.scalar(); m->ior = k.scalar();
				sc->materials.emplace(m->name, m);
			}
			if (what == "shape") {
				std::string kind = k.word();
				if (kind == "box") {  // scene.cpp:43-54
					std::string n = k.word();
					V3 mn = k.vec(), mx = k.vec();
					try {
						sc->root->add_child(std::make_shared<Box>(mn, mx, n, find_material(*sc, k.word())));
					} catch (const char* e) { sc->error = e; }
				} else if (kind == "sphere") {  // scene.cpp:56-68
					std::string n = k.word();
					V3 c = k.vec();
					float r = k.scalar();
					sc->root->add_child(std::make_shared<Sphere>(r, c, n, find_material(*sc, k.word())));
				} else if (kind == "triangle") {  // scene.cpp:70-82
					std::string n = k.word();
					V3 a = k.vec(), b = k.vec(), c = k.vec();
					sc->root->add_child(std::make_shared<Triangle>(a, b, c, n, find_material(*sc, k.word())));
				} else if (kind == "obj") {  // scene.cpp:323-326
					sc->root->add_child(load_obj(obj_dir, k.word(), sc->error));
				}
			} else if (what == "light") {  // scene.cpp:84-93
				k.word();
				V3 pos = k.vec();
				Color col = k.color();
				float br = k.scalar();
				sc->lights.push_back({pos, col * br});
			} else if (what == "ambient") {  // scene.cpp:95-105: read order r, B, G
				k.word();
				Color col;
				col.r = k.member(); col.b = k.member(); col.g = k.member();
				float br = k.scalar();
				sc->ambient = col * br;
			} else if (what == "camera") {  // scene.cpp:107-126
				k.word();
				Camera cam;
				cam.fov_x = k.scalar();
				cam.position.x = k.member(); cam.position.y = k.member(); cam.position.z = k.member();
				float yaw = k.scalar(), pitch = k.scalar(), roll = k.scalar();
				M4 rot = euler_yxz(radians(yaw), radians(pitch), radians(roll));
				cam.direction = xform({0, 0, -1}, rot, false);
				cam.up = xform({0, 1, 0}, rot, false);
				sc->camera = cam;
			}
		} else if (tok == "transform") {  // scene.cpp:337-373
			std::string n = k.word(), action = k.word();
			auto it = sc->root->children.find(n);
			if (it == sc->root->children.end()) { sc->error = "transform of unknown shape '" + n + "'"; continue; }
			if (action == "translate") {
				float x = k.scalar(), y = k.scalar(), z = k.scalar();
				it->second->do_translate(x, y, z);
			} else if (action == "rotate") {
				float a = k.scalar(), b = k.scalar(), c = k.scalar();
				it->second->do_rotate(radians(a), radians(b), radians(c));
			} else if (action == "scale") {
				float a = k.scalar(), b = k.scalar(), c = k.scalar();
				it->second->do_scale(a, b, c);
			}
		}
		// "render ..." is parsed and ignored (scene.cpp:293-305)
	}
	sc->root->build_octree();  // scene.cpp:398
	return sc;
}

void rank_leaves(Scene& sc, Composite const& c, std::string const& prefix) {
	for (auto const& it : c.children) {
		Shape const* s = it.second.get();
		if (auto const* sub = dynamic_cast<Composite const*>(s)) {
			rank_leaves(sc, *sub, prefix + sub->name + "/");
		} else if (!sc.rank.count(s)) {
			sc.rank[s] = (int) sc.leaves.size();
			sc.leaves.push_back(s);
			sc.leaf_path.push_back(prefix + s->name);
		}
	}
}

// ---------------------------------------------------------------------------
// Renderer (framework/renderer.cpp)
// ---------------------------------------------------------------------------
struct Renderer {
	Scene const& sc;
	unsigned w, h, aa, max_bounces;

	// renderer.cpp:252-276
	float schlick(V3 rd, V3 n, float ior) const {
		float n1 = 1, n2 = ior;
		float cos_in = -dot(n, rd);
		if (cos_in < 0) std::swap(n1, n2);
		if (n1 > n2) {
			float eta = n1 / n2;
			float s2 = eta * eta * (1 - cos_in * cos_in);
			if (s2 >= 1) return 1;
			cos_in = sqrtf(1 - s2);
		}
		float r0 = (1 - ior) / (1 + ior);
		r0 *= r0;
		float f = 1 - cos_in;
		return r0 + (1 - r0) * f * f * f * f * f;
	}
	// renderer.cpp:177-192 — pow resolves to the double overload (operands promoted)
	Color specular(V3 view, V3 l, V3 n, Color I, Material const& m) const {
		V3 r = 2 * dot(n, l) * n - l;
		float c = dot(r, view * -1.0f);
		if (c < 0) return {};
		float f = (float) ::pow((double) c, (double) m.m);
		return I * m.ks * f;
	}
	// renderer.cpp:137-175
	Color phong(Hit const& hp) const {
		Material const& m = *hp.leaf->material;
		Color col = sc.ambient * m.ka;
		for (PointLight const& L : sc.lights) {
			V3 ld = L.position - hp.position;
			float dist = length(ld);
			ld = normalize(ld);
			if (tl_counters) ++tl_counters->shadow_rays;
			Hit blk = sc.root->intersect(Ray{hp.position, ld});
			if (blk.does && blk.distance <= dist) continue;
			float c = dot(hp.normal, ld);
			if (c < 0) continue;
			col += L.intensity * m.kd * c;
			if (m.m != 0) col += specular(hp.ray_direction, ld, hp.normal, L.intensity, m);
		}
		return col;
	}
	// renderer.cpp:194-201
	Color reflection(Hit const& hp, unsigned depth) const {
		float c = -dot(hp.normal, hp.ray_direction);
		V3 nd = hp.ray_direction + (hp.normal * c * 2.0f);
		return trace(Ray{hp.position, nd}, depth + 1) * hp.leaf->material->ks;
	}
	// renderer.cpp:203-225
	Color refraction(Hit const& hp, unsigned depth) const {
		V3 rd = hp.ray_direction, n = hp.normal;
		float eta = 1 / hp.leaf->material->ior;
		float c = -dot(n, rd);
		if (c < 0) {
			eta = 1 / eta;
			c *= -1;
			n = n * -1.0f;
		}
		float k = 1 - eta * eta * (1 - c * c);
		if (k < 0) return reflection(hp, depth);
		V3 nd = rd * eta + n * (eta * c - sqrtf(k));
		return trace(Ray{hp.position - n * (2 * kEps), nd}, depth + 1) * hp.leaf->material->kd;
	}
	// renderer.cpp:113-135
	Color shade(Hit const& hp, unsigned depth) const {
		if (tl_counters) ++tl_counters->shaded_hits;
		Material const& m = *hp.leaf->material;
		Color c = phong(hp) * m.opacity;
		if (depth >= max_bounces) return c;
		if (m.glossy > 0 && m.opacity < 1) {
			float R = schlick(hp.ray_direction, hp.normal, m.ior);
			c *= R * m.opacity;
			c += reflection(hp, depth) * R;
			c += refraction(hp, depth) * (1 - R) * (1 - m.opacity);
		} else if (m.glossy > 0) {
			float R = schlick(hp.ray_direction, hp.normal, m.ior);
			R = m.glossy + (1 - m.glossy) * R;
			c *= 1 - R;
			c += reflection(hp, depth) * R;
		} else if (m.opacity < 1) {
			c *= m.opacity;
			c += refraction(hp, depth) * (1 - m.opacity);
		}
		return c;
	}
	// renderer.cpp:108-111
	Color trace(Ray const& r, unsigned depth = 0) const {
		if (tl_counters) ++tl_counters->trace_rays;
		Hit hp = sc.root->intersect(r);
		return hp.does ? shade(hp, depth) : Color{};
	}
};

struct CameraBasis {
	float dist;
	M4 m;
};
// renderer.cpp:27-37.  tan() is the double overload; the division is done in double and
// rounded once on assignment.
CameraBasis camera_basis(Camera const& cam, unsigned w) {
	float fov_rad = cam.fov_x / 180 * 3.14159265f;
	float dist = (w / 2.0f) / ::tan((double) (fov_rad / 2));
	V3 u = cross(cam.direction, cam.up);
	V3 v = cross(u, cam.direction);
	V3 nd = neg(cam.direction);
	M4 m{{{u.x, u.y, u.z, 0}, {v.x, v.y, v.z, 0}, {nd.x, nd.y, nd.z, 0}, {cam.position.x, cam.position.y, cam.position.z, 1}}};
	return {dist, m};
}
// renderer.cpp:76-82
Ray primary_ray(CameraBasis const& cb, unsigned w, unsigned h, unsigned x, unsigned y, int aax, int aay, float aa_unit) {
	V3 pp{x + aax * aa_unit - (w * 0.5f), y + aay * aa_unit - (h * 0.5f), -cb.dist};
	V3 nn = normalize(pp);
	V4 d = mul(cb.m, V4{nn.x, nn.y, nn.z, 0});
	return Ray{{cb.m.c[3].x, cb.m.c[3].y, cb.m.c[3].z}, {d.x, d.y, d.z}};
}

// Jittered sampling - NOT in the reference (which samples the corner of every sub-pixel cell
// and has no random numbers): the product offers it as an option, so the checker restates the
// product's published generator (include/ptx.h, ptx_jitter) to be able to check that mode too.
// Parity of this mode is pinned by nothing in the reference.
inline uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
inline void jitter_offsets(uint32_t seed, uint32_t pixel, uint32_t sample, float& dx, float& dy) {
	uint32_t k = mix32(seed) ^ mix32(pixel * 0x9e3779b9u + 0x7f4a7c15u) ^ mix32(sample * 0x85ebca6bu + 0xc2b2ae35u);
	uint32_t a = mix32(k), b = mix32(k ^ 0x68e31da4u);
	dx = (float) (a >> 8) * 5.9604644775390625e-08f;
	dy = (float) (b >> 8) * 5.9604644775390625e-08f;
}
Ray jittered_ray(CameraBasis const& cb, unsigned w, unsigned h, unsigned x, unsigned y, int aax, int aay, unsigned aa, float aa_unit, uint32_t seed) {
	float jx, jy;
	jitter_offsets(seed, y * w + x, (uint32_t) (aax * (int) aa + aay), jx, jy);
	V3 pp{x + (aax + jx) * aa_unit - (w * 0.5f), y + (aay + jy) * aa_unit - (h * 0.5f), -cb.dist};
	V3 nn = normalize(pp);
	V4 d = mul(cb.m, V4{nn.x, nn.y, nn.z, 0});
	return Ray{{cb.m.c[3].x, cb.m.c[3].y, cb.m.c[3].z}, {d.x, d.y, d.z}};
}

// ppmwriter.cpp:41-43: (unsigned) max(0.0, min(255.0 * c, 255.0)) evaluated in double
inline unsigned char quantise(float c) {
	double v = std::min(255.0 * c, 255.0);
	v = std::max(0.0, v);
	return (unsigned char) (unsigned int) v;
}

}  // namespace

// ===========================================================================
// C interface for ctypes (tests, golden generation, bench cpu_baseline)
// ===========================================================================
extern "C" {

void* oracle_load_scene(const char* sdf_path, const char* obj_dir) {
	auto sc = load_scene(sdf_path, obj_dir);
	rank_leaves(*sc, *sc->root, "");
	return sc.release();
}
void oracle_free_scene(void* s) { delete static_cast<Scene*>(s); }
// A Composite of a caller's own with spheres and boxes in it, added to the root (composite.hpp:9-29, sphere.hpp:10, box.hpp:9):
// kinds[i] 0 = sphere (cx cy cz r), 1 = box (min, max), 7 floats per child; names and material names joined by ';'
int oracle_add_group(void* s, const char* name, int n, const int* kinds, const float* params, const char* child_names, const char* material_names) {
	Scene* sc = static_cast<Scene*>(s);
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
	std::vector<std::shared_ptr<Composite>> above{sc->root};
	std::string own(name ? name : "");
	for (size_t cut; (cut = own.find('/')) != std::string::npos; own = own.substr(cut + 1)) {
		auto it = above.back()->children.find(own.substr(0, cut));
		if (it == above.back()->children.end()) return 1;
		auto next = std::dynamic_pointer_cast<Composite>(it->second);
		if (!next) return 1;
		above.push_back(next);
	}
	auto g = std::make_shared<Composite>(own);
	for (int i = 0; i < n; ++i) {
		const float* p = params + 7 * i;
		auto it = sc->materials.find(mats[(size_t) i]);
		if (it == sc->materials.end()) return 1;
		if (kinds[i] == 0) g->add_child(std::make_shared<Sphere>(p[3], V3{p[0], p[1], p[2]}, names[(size_t) i], it->second));
		else g->add_child(std::make_shared<Box>(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, names[(size_t) i], it->second));
	}
	g->build_octree();
	above.back()->add_child(g);
	for (size_t k = above.size(); k-- > 0;) above[k]->build_octree();
	sc->rank.clear();
	sc->leaves.clear();
	sc->leaf_path.clear();
	rank_leaves(*sc, *sc->root, "");
	return 0;
}
// root->find_child(a)->find_child(b)...->translate / rotate / scale, as a C++ caller of the reference moves a part of a
// loaded mesh (composite.cpp:112-114, :168-181): path levels separated by '/', op 1 translate, 2 rotate (radians), 3 scale
int oracle_transform_child(void* s, const char* path, int op, float a, float b, float c) {
	Scene* sc = static_cast<Scene*>(s);
	std::shared_ptr<Shape> shape = sc->root;
	std::string rest(path ? path : "");
	while (!rest.empty()) {
		const size_t cut = rest.find('/');
		const std::string name = rest.substr(0, cut);
		rest = cut == std::string::npos ? std::string() : rest.substr(cut + 1);
		auto* comp = dynamic_cast<Composite*>(shape.get());
		if (!comp) return 1;
		auto it = comp->children.find(name);
		if (it == comp->children.end()) return 1;
		shape = it->second;
	}
	if (op == 1) shape->do_translate(a, b, c);
	else if (op == 2) shape->do_rotate(a, b, c);
	else if (op == 3) shape->do_scale(a, b, c);
	else return 1;
	sc->rank.clear();
	sc->leaves.clear();
	sc->leaf_path.clear();
	rank_leaves(*sc, *sc->root, "");
	return 0;
}
const char* oracle_scene_error(void* s) { return static_cast<Scene*>(s)->error.c_str(); }
int oracle_leaf_count(void* s) { return (int) static_cast<Scene*>(s)->leaves.size(); }
// leaf name (as HitPoint::hit_object would carry it) and slash-joined path of rank i
const char* oracle_leaf_name(void* s, int i) { return static_cast<Scene*>(s)->leaves[i]->name.c_str(); }
const char* oracle_leaf_path(void* s, int i) { return static_cast<Scene*>(s)->leaf_path[i].c_str(); }

void oracle_scene_info(void* s, float* cam10, float* ambient3, int* n_lights, float* lights, float* root_min3, float* root_max3) {
	Scene* sc = static_cast<Scene*>(s);
	cam10[0] = sc->camera.fov_x;
	float const* p = &sc->camera.position.x; float const* d = &sc->camera.direction.x; float const* u = &sc->camera.up.x;
	for (int c = 0; c < 3; ++c) { cam10[1 + c] = p[c]; cam10[4 + c] = d[c]; cam10[7 + c] = u[c]; }
	ambient3[0] = sc->ambient.r; ambient3[1] = sc->ambient.g; ambient3[2] = sc->ambient.b;
	*n_lights = (int) sc->lights.size();
	for (int i = 0; i < *n_lights && i < 16; ++i) {
		lights[6 * i + 0] = sc->lights[i].position.x; lights[6 * i + 1] = sc->lights[i].position.y; lights[6 * i + 2] = sc->lights[i].position.z;
		lights[6 * i + 3] = sc->lights[i].intensity.r; lights[6 * i + 4] = sc->lights[i].intensity.g; lights[6 * i + 5] = sc->lights[i].intensity.b;
	}
	V3 mn = sc->root->bmin(identity()), mx = sc->root->bmax(identity());
	root_min3[0] = mn.x; root_min3[1] = mn.y; root_min3[2] = mn.z;
	root_max3[0] = mx.x; root_max3[1] = mx.y; root_max3[2] = mx.z;
}

// Renderer::render + thread_function (renderer.cpp:25-91) + write (:93-106) + PpmWriter::write
// (ppmwriter.cpp:33-45).  color_buffer: w*h*3 tone-mapped floats, row-major y-up.
// ppm_rgb: w*h*3 bytes, rows flipped as the PPM stores them.  counters (optional): 7 x u64
// {trace rays, shadow rays, shaded hits, sphere tests, box tests, triangle tests, bounds tests}.
// threads <= 0 => hardware_concurrency (renderer.cpp:39).  Returns wall seconds of the pixel loop.
double oracle_render_jittered(void* s, unsigned w, unsigned h, unsigned aa, unsigned bounces, int threads, float* color_buffer,
                              unsigned char* ppm_rgb, uint64_t* counters, uint32_t jitter_seed);
double oracle_render(void* s, unsigned w, unsigned h, unsigned aa, unsigned bounces, int threads,
                     float* color_buffer, unsigned char* ppm_rgb, uint64_t* counters) {
	return oracle_render_jittered(s, w, h, aa, bounces, threads, color_buffer, ppm_rgb, counters, 0);
}
// the same loop; jitter_seed != 0 moves every sample inside its cell (jittered_ray above)
double oracle_render_jittered(void* s, unsigned w, unsigned h, unsigned aa, unsigned bounces, int threads, float* color_buffer,
                              unsigned char* ppm_rgb, uint64_t* counters, uint32_t jitter_seed) {
	Scene* sc = static_cast<Scene*>(s);
	Renderer R{*sc, w, h, aa, bounces};
	CameraBasis cb = camera_basis(sc->camera, w);
	size_t nthreads = threads > 0 ? (size_t) threads : std::thread::hardware_concurrency();
	if (nthreads == 0) nthreads = 1;
	std::atomic_uint pixel_index{0};
	std::vector<Counters> cnt(nthreads);
	auto worker = [&](size_t tid) {
		tl_counters = counters ? &cnt[tid] : nullptr;
		float aa_unit = 1.0f / aa;
		while (true) {
			unsigned cur = pixel_index++;
			unsigned x = cur % w, y = cur / w;
			if (cur >= w * h) break;
			Color acc{};
			for (int aax = 0; aax < (int) aa; ++aax)
				for (int aay = 0; aay < (int) aa; ++aay)
					acc += R.trace(jitter_seed ? jittered_ray(cb, w, h, x, y, aax, aay, aa, aa_unit, jitter_seed)
					                           : primary_ray(cb, w, h, x, y, aax, aay, aa_unit));
			acc *= 1.0f / (aa * aa);
			// tone map, renderer.cpp:286-291
			acc.r /= acc.r + 1; acc.g /= acc.g + 1; acc.b /= acc.b + 1;
			size_t pos = (size_t) w * y + x;
			if (color_buffer) { color_buffer[3 * pos] = acc.r; color_buffer[3 * pos + 1] = acc.g; color_buffer[3 * pos + 2] = acc.b; }
			if (ppm_rgb) {
				size_t q = 3 * ((size_t) w * (h - 1 - y) + x);
				ppm_rgb[q] = quantise(acc.r); ppm_rgb[q + 1] = quantise(acc.g); ppm_rgb[q + 2] = quantise(acc.b);
			}
		}
		tl_counters = nullptr;
	};
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> pool;
	for (size_t i = 0; i < nthreads; ++i) pool.emplace_back(worker, i);
	for (auto& t : pool) t.join();
	auto t1 = std::chrono::steady_clock::now();
	if (counters) {
		Counters sum;
		for (auto const& c : cnt) {
			sum.trace_rays += c.trace_rays; sum.shadow_rays += c.shadow_rays; sum.shaded_hits += c.shaded_hits;
			sum.sphere_tests += c.sphere_tests; sum.box_tests += c.box_tests; sum.tri_tests += c.tri_tests; sum.bounds_tests += c.bounds_tests;
		}
		counters[0] = sum.trace_rays; counters[1] = sum.shadow_rays; counters[2] = sum.shaded_hits;
		counters[3] = sum.sphere_tests; counters[4] = sum.box_tests; counters[5] = sum.tri_tests; counters[6] = sum.bounds_tests;
	}
	return std::chrono::duration<double>(t1 - t0).count();
}

// Un-tone-mapped per-SAMPLE colours (trace() of every primary ray, pixel-major, aax outer),
// for tests that want to look below the per-pixel average.
void oracle_sample_colors(void* s, unsigned w, unsigned h, unsigned aa, unsigned bounces, unsigned x0, unsigned y0,
                          unsigned nx, unsigned ny, float* out) {
	Scene* sc = static_cast<Scene*>(s);
	Renderer R{*sc, w, h, aa, bounces};
	CameraBasis cb = camera_basis(sc->camera, w);
	float aa_unit = 1.0f / aa;
	// every sample stands alone, so the rows of the sample grid (one pixel, one aax) are dealt to the host's threads
	const size_t rows = (size_t) nx * ny * aa;
	size_t nthreads = std::max(1u, std::thread::hardware_concurrency());
	if (const char* e = std::getenv("OMP_NUM_THREADS"))  // the share of the machine this process was given, where one is set
		if (std::atoi(e) > 0) nthreads = std::min<size_t>(nthreads, (size_t) std::atoi(e));
	nthreads = std::min<size_t>(nthreads, std::min<size_t>(rows, 64));
	std::atomic<size_t> next{0};
	auto worker = [&]() {
		for (size_t r = next++; r < rows; r = next++) {
			const unsigned pix = (unsigned) (r / aa);
			const int aax = (int) (r % aa);
			const unsigned x = x0 + pix % nx, y = y0 + pix / nx;
			float* o = out + r * aa * 3;
			for (int aay = 0; aay < (int) aa; ++aay) {
				Color c = R.trace(primary_ray(cb, w, h, x, y, aax, aay, aa_unit));
				*o++ = c.r; *o++ = c.g; *o++ = c.b;
			}
		}
	};
	std::vector<std::thread> pool;
	for (size_t i = 1; i < nthreads; ++i) pool.emplace_back(worker);
	worker();
	for (auto& t : pool) t.join();
}

// PpmWriter::save (ppmwriter.cpp:55-76): "P3 W H 255 \n", value + ' ', '\n' after every 19th.
// Returns the number of bytes written to `out` (capacity `cap`), or the needed size if too small.
long oracle_ppm_text(const unsigned char* rgb, unsigned w, unsigned h, char* out, long cap) {
	std::string sres = "P3 " + std::to_string(w) + " " + std::to_string(h) + " 255 \n";
	sres.reserve((size_t) w * h * 3 * 4 + 64);
	size_t pos = 1;
	size_t n = (size_t) w * h * 3;
	char buf[8];
	for (size_t i = 0; i < n; ++i, ++pos) {
		int len = std::snprintf(buf, sizeof buf, "%d ", (int) rgb[i]);
		sres.append(buf, (size_t) len);
		if (pos > 18) { sres.push_back('\n'); pos = 0; }
	}
	if ((long) sres.size() <= cap) std::memcpy(out, sres.data(), sres.size());
	return (long) sres.size();
}

void oracle_primary_rays(void* s, unsigned w, unsigned h, unsigned aa, float* origins, float* dirs) {
	Scene* sc = static_cast<Scene*>(s);
	CameraBasis cb = camera_basis(sc->camera, w);
	float aa_unit = 1.0f / aa;
	size_t k = 0;
	for (unsigned y = 0; y < h; ++y)
		for (unsigned x = 0; x < w; ++x)
			for (int aax = 0; aax < (int) aa; ++aax)
				for (int aay = 0; aay < (int) aa; ++aay) {
					Ray r = primary_ray(cb, w, h, x, y, aax, aay, aa_unit);
					origins[3 * k] = r.o.x; origins[3 * k + 1] = r.o.y; origins[3 * k + 2] = r.o.z;
					dirs[3 * k] = r.d.x; dirs[3 * k + 1] = r.d.y; dirs[3 * k + 2] = r.d.z;
					++k;
				}
}

// scene.root->intersect for n rays (renderer.cpp:109).  leaf_rank: first depth-first
// encounter rank of the hit leaf, or -1.
void oracle_intersect(void* s, long n, const float* o, const float* d, int* leaf_rank, float* t, float* pos, float* nrm, float* rdir) {
	Scene* sc = static_cast<Scene*>(s);
	for (long i = 0; i < n; ++i) {
		Ray r{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
		Hit hp = sc->root->intersect(r);
		leaf_rank[i] = hp.does ? sc->rank.at(hp.leaf) : -1;
		t[i] = hp.distance;
		float const* P = &hp.position.x; float const* N = &hp.normal.x; float const* D = &hp.ray_direction.x;
		for (int c = 0; c < 3; ++c) { pos[3 * i + c] = P[c]; nrm[3 * i + c] = N[c]; rdir[3 * i + c] = D[c]; }
	}
}

// single primitives, as source/tests.cpp constructs them (kind 0 sphere{c,r} 1 box{min,max}
// 2 triangle{v0,v1,v2}); xf = nxf x {op(1 translate,2 rotate[rad],3 scale), a, b, c}
int oracle_prim_intersect(int kind, const float* p, const float* xf, int nxf, long n, const float* o, const float* d,
                          int* hit, float* t, float* pos, float* nrm, float* rdir) {
	auto mat = std::make_shared<Material>();
	std::shared_ptr<Shape> sh;
	try {
		if (kind == 0) sh = std::make_shared<Sphere>(p[3], V3{p[0], p[1], p[2]}, "s", mat);
		else if (kind == 1) sh = std::make_shared<Box>(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, "b", mat);
		else sh = std::make_shared<Triangle>(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, V3{p[6], p[7], p[8]}, "t", mat);
	} catch (const char*) { return -1; }
	for (int k = 0; k < nxf; ++k) {
		const float* q = xf + 4 * k;
		if (q[0] == 1) sh->do_translate(q[1], q[2], q[3]);
		else if (q[0] == 2) sh->do_rotate(q[1], q[2], q[3]);
		else if (q[0] == 3) sh->do_scale(q[1], q[2], q[3]);
	}
	for (long i = 0; i < n; ++i) {
		Ray r{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
		Hit hp = sh->intersect(r);
		hit[i] = hp.does;
		t[i] = hp.distance;
		float const* P = &hp.position.x; float const* N = &hp.normal.x; float const* D = &hp.ray_direction.x;
		for (int c = 0; c < 3; ++c) { pos[3 * i + c] = P[c]; nrm[3 * i + c] = N[c]; rdir[3 * i + c] = D[c]; }
	}
	return 0;
}

// bounds of a transformed box, for the rotated-box KAT of source/tests.cpp:235-250
void oracle_box_bounds(const float* p, const float* xf, int nxf, float* mn3, float* mx3) {
	Box b(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]});
	for (int k = 0; k < nxf; ++k) {
		const float* q = xf + 4 * k;
		if (q[0] == 1) b.do_translate(q[1], q[2], q[3]);
		else if (q[0] == 2) b.do_rotate(q[1], q[2], q[3]);
		else if (q[0] == 3) b.do_scale(q[1], q[2], q[3]);
	}
	V3 mn = b.bmin(identity()), mx = b.bmax(identity());
	mn3[0] = mn.x; mn3[1] = mn.y; mn3[2] = mn.z;
	mx3[0] = mx.x; mx3[1] = mx.y; mx3[2] = mx.z;
}

}  // extern "C"
