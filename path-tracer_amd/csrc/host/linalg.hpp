// Minimal vector/matrix types for the host side of the render path.  The reference uses
// GLM 0.9.5.3 for its scene graph; what is needed of it is restated here with the SAME
// evaluation order, because transforms, inverses and bounds computed at load time feed
// the per-ray arithmetic on the GPU and have to carry the same bits
// (glm/detail/type_mat4x4.inl:476-535, :651-687, :753-775; gtc/matrix_transform.inl:36-45,
// :149-161; gtx/euler_angles.inl:119-151; detail/func_geometric.inl; func_common.inl:365-400).
#pragma once
#include <cmath>

struct vec3 {
	float x = 0, y = 0, z = 0;
	vec3() = default;
	vec3(float a, float b, float c) : x(a), y(b), z(c) {}
	float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
inline vec3 operator+(vec3 const& a, vec3 const& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 const& a, vec3 const& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 const& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator-(vec3 const& a) { return {-a.x, -a.y, -a.z}; }
inline float dot(vec3 const& a, vec3 const& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 const& a, vec3 const& b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 const& v) { return v * (1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z)); }
inline vec3 min3(vec3 const& a, vec3 const& b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline vec3 max3(vec3 const& a, vec3 const& b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }

struct vec4 {
	float x = 0, y = 0, z = 0, w = 0;
};
inline vec4 operator*(vec4 const& a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(vec4 const& a, vec4 const& b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator+(vec4 const& a, vec4 const& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 const& a, vec4 const& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

// column-major like glm::mat4; the memory of `col` is the 16 floats the C ABI takes
struct mat4 {
	vec4 col[4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
	float at(int c, int r) const { return r == 0 ? col[c].x : r == 1 ? col[c].y : r == 2 ? col[c].z : col[c].w; }
	const float* data() const { return &col[0].x; }
};
static_assert(sizeof(mat4) == 64, "mat4 must be 16 packed floats");

inline vec4 operator*(mat4 const& m, vec4 const& v) { return (m.col[0] * v.x + m.col[1] * v.y) + (m.col[2] * v.z + m.col[3] * v.w); }
inline mat4 operator*(mat4 const& a, mat4 const& b) {
	mat4 r;
	for (int j = 0; j < 4; ++j) r.col[j] = a.col[0] * b.col[j].x + a.col[1] * b.col[j].y + a.col[2] * b.col[j].z + a.col[3] * b.col[j].w;
	return r;
}
inline mat4 translate(mat4 const& m, vec3 const& v) {
	mat4 r = m;
	r.col[3] = m.col[0] * v.x + m.col[1] * v.y + m.col[2] * v.z + m.col[3];
	return r;
}
inline mat4 scale(mat4 const& m, vec3 const& v) {
	mat4 r;
	r.col[0] = m.col[0] * v.x;
	r.col[1] = m.col[1] * v.y;
	r.col[2] = m.col[2] * v.z;
	r.col[3] = m.col[3];
	return r;
}
inline mat4 euler_angle_yxz(float yaw, float pitch, float roll) {
	float ch = std::cos(yaw), sh = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cb = std::cos(roll), sb = std::sin(roll);
	mat4 r;
	r.col[0] = {ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0};
	r.col[1] = {-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0};
	r.col[2] = {sh * cp, -sp, ch * cp, 0};
	r.col[3] = {0, 0, 0, 1};
	return r;
}
inline float radians(float degrees) { return degrees * float(0.01745329251994329576923690768489); }

inline mat4 inverse(mat4 const& m) {
	float c00 = m.at(2, 2) * m.at(3, 3) - m.at(3, 2) * m.at(2, 3);
	float c02 = m.at(1, 2) * m.at(3, 3) - m.at(3, 2) * m.at(1, 3);
	float c03 = m.at(1, 2) * m.at(2, 3) - m.at(2, 2) * m.at(1, 3);
	float c04 = m.at(2, 1) * m.at(3, 3) - m.at(3, 1) * m.at(2, 3);
	float c06 = m.at(1, 1) * m.at(3, 3) - m.at(3, 1) * m.at(1, 3);
	float c07 = m.at(1, 1) * m.at(2, 3) - m.at(2, 1) * m.at(1, 3);
	float c08 = m.at(2, 1) * m.at(3, 2) - m.at(3, 1) * m.at(2, 2);
	float c10 = m.at(1, 1) * m.at(3, 2) - m.at(3, 1) * m.at(1, 2);
	float c11 = m.at(1, 1) * m.at(2, 2) - m.at(2, 1) * m.at(1, 2);
	float c12 = m.at(2, 0) * m.at(3, 3) - m.at(3, 0) * m.at(2, 3);
	float c14 = m.at(1, 0) * m.at(3, 3) - m.at(3, 0) * m.at(1, 3);
	float c15 = m.at(1, 0) * m.at(2, 3) - m.at(2, 0) * m.at(1, 3);
	float c16 = m.at(2, 0) * m.at(3, 2) - m.at(3, 0) * m.at(2, 2);
	float c18 = m.at(1, 0) * m.at(3, 2) - m.at(3, 0) * m.at(1, 2);
	float c19 = m.at(1, 0) * m.at(2, 2) - m.at(2, 0) * m.at(1, 2);
	float c20 = m.at(2, 0) * m.at(3, 1) - m.at(3, 0) * m.at(2, 1);
	float c22 = m.at(1, 0) * m.at(3, 1) - m.at(3, 0) * m.at(1, 1);
	float c23 = m.at(1, 0) * m.at(2, 1) - m.at(2, 0) * m.at(1, 1);
	vec4 f0{c00, c00, c02, c03}, f1{c04, c04, c06, c07}, f2{c08, c08, c10, c11};
	vec4 f3{c12, c12, c14, c15}, f4{c16, c16, c18, c19}, f5{c20, c20, c22, c23};
	vec4 v0{m.at(1, 0), m.at(0, 0), m.at(0, 0), m.at(0, 0)};
	vec4 v1{m.at(1, 1), m.at(0, 1), m.at(0, 1), m.at(0, 1)};
	vec4 v2{m.at(1, 2), m.at(0, 2), m.at(0, 2), m.at(0, 2)};
	vec4 v3{m.at(1, 3), m.at(0, 3), m.at(0, 3), m.at(0, 3)};
	vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
	vec4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
	vec4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
	vec4 i3 = v0 * f2 - v1 * f4 + v2 * f5;
	vec4 sa{+1, -1, +1, -1}, sb{-1, +1, -1, +1};
	mat4 inv;
	inv.col[0] = i0 * sa;
	inv.col[1] = i1 * sb;
	inv.col[2] = i2 * sa;
	inv.col[3] = i3 * sb;
	vec4 row0{inv.col[0].x, inv.col[1].x, inv.col[2].x, inv.col[3].x};
	vec4 d0 = m.col[0] * row0;
	float det = (d0.x + d0.y) + (d0.z + d0.w);
	float ood = 1.0f / det;
	for (auto& c : inv.col) c = c * ood;
	return inv;
}

// transform_vec of framework/shape.cpp:46-48
inline vec3 transform_vec(vec3 const& v, mat4 const& m, bool is_location = true) {
	vec4 r = m * vec4{v.x, v.y, v.z, is_location ? 1.0f : 0.0f};
	return {r.x, r.y, r.z};
}
