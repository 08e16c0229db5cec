// Device-side arithmetic of the render path for gfx950, written so that every value the
// reference computes per ray (hit distance, position, normal, local ray direction) comes
// out with the same fp32 operation sequence: this translation unit is compiled with
// -ffp-contract=off (no FMA contraction; the reference build has none), IEEE divide and
// sqrt, and the min/max selections are the C++ ternaries of std::min/std::max/glm::min,
// not fminf/fmaxf (which differ on NaN).  The ONLY place that may round differently is
// the BVH node test (bvh_slab), which is a conservative filter and never produces a value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptxd {

constexpr float kEps = 0.001f;  // EPSILON of framework/{renderer,sphere,box,triangle}.cpp
constexpr float kInf = __builtin_huge_valf();

// (the plain vector helpers are also callable on the host: the scene upload precomputes the
// normals that do not depend on the ray with these very functions, same operations, same order)
struct f3 {
	float x, y, z;
};
__host__ __device__ __forceinline__ f3 mk(float x, float y, float z) { return {x, y, z}; }
__host__ __device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__host__ __device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__host__ __device__ __forceinline__ f3 cmul(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // Color * Color
// glm detail/func_geometric.inl:61-68, :212-224, :104-110, :252-261
__host__ __device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
__host__ __device__ __forceinline__ float length(f3 v) { return __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
__host__ __device__ __forceinline__ f3 normalize(f3 v) {
	float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
	return v * (1.0f / __builtin_sqrtf(sqr));
}
// What one Composite with an identity transform (the root, an octant, an OBJ sub-object) does to the normal of a
// hit it passes up (framework/composite.cpp:100): normalize(transform_vec(n, identity, false)).  With the identity
// matrix, glm's (m0*x + m1*y) + (m2*z + m3*0) leaves a finite component as it is, except that a negative zero comes
// out as positive zero - which is exactly what adding +0.0f does.
__host__ __device__ __forceinline__ f3 up_one_level(f3 n) { return normalize(f3{n.x + 0.0f, n.y + 0.0f, n.z + 0.0f}); }
__host__ __device__ __forceinline__ bool any_nonzero(f3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }

// std::min / std::max semantics (b < a ? b : a  /  a < b ? b : a), framework/box.cpp:123-147
__device__ __forceinline__ float smin(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float smax(float a, float b) { return a < b ? b : a; }

// glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*w), detail/type_mat4x4.inl:676-687.
// M is column-major (element [c*4+r]).  transform_vec of framework/shape.cpp:46-48.
__host__ __device__ __forceinline__ f3 xform_point(const float* __restrict__ M, f3 v) {
	return {(M[0] * v.x + M[4] * v.y) + (M[8] * v.z + M[12] * 1.0f), (M[1] * v.x + M[5] * v.y) + (M[9] * v.z + M[13] * 1.0f),
	        (M[2] * v.x + M[6] * v.y) + (M[10] * v.z + M[14] * 1.0f)};
}
__host__ __device__ __forceinline__ f3 xform_dir(const float* __restrict__ M, f3 v) {
	return {(M[0] * v.x + M[4] * v.y) + (M[8] * v.z + M[12] * 0.0f), (M[1] * v.x + M[5] * v.y) + (M[9] * v.z + M[13] * 0.0f),
	        (M[2] * v.x + M[6] * v.y) + (M[10] * v.z + M[14] * 0.0f)};
}

// Counter-based generator for jittered sampling: two 24-bit uniform numbers in [0, 1) from
// (seed, pixel, sample) through an integer finaliser (xor-shift-multiply rounds).  No state:
// every sample's offsets can be formed anywhere, in any order.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
	x ^= x >> 16;
	x *= 0x7feb352du;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}
__host__ __device__ __forceinline__ void jitter_offsets(uint32_t seed, uint32_t pixel, uint32_t sample, float& dx, float& dy) {
	uint32_t k = mix32(seed) ^ mix32(pixel * 0x9e3779b9u + 0x7f4a7c15u) ^ mix32(sample * 0x85ebca6bu + 0xc2b2ae35u);
	uint32_t a = mix32(k), b = mix32(k ^ 0x68e31da4u);
	dx = (float) (a >> 8) * 5.9604644775390625e-08f;  // 2^-24: exact, < 1
	dy = (float) (b >> 8) * 5.9604644775390625e-08f;
}

// The intersectors below are written WITHOUT early returns: every lane runs the whole
// formula and the reference's `return false` conditions are OR-ed into one flag.  The
// values that survive are computed by the same operations in the same order, so results
// are unchanged; what changes is that a wave executes one straight block per primitive
// instead of a ladder of exec-mask branches (a rejected lane's leftovers - inf, NaN from
// 1/0 or sqrt(<0) - are never read).

// Box::intersect(Ray, float&) of framework/box.cpp:105-165 on an already-local ray, with the
// three reciprocals 1/d supplied (the reference forms them inside, :119, :132, :143).
__device__ __forceinline__ bool box_slab_inv(f3 lo, f3 hi, f3 o, f3 d, f3 inv, float& t) {
	const bool zx = (0.0f == d.x), zy = (0.0f == d.y), zz = (0.0f == d.z);
	float t_min = -kInf, t_max = kInf;
	{
		float t1 = (lo.x - o.x) * inv.x, t2 = (hi.x - o.x) * inv.x;
		t_min = zx ? t_min : smax(t_min, smin(t1, t2));
		t_max = zx ? t_max : smin(t_max, smax(t1, t2));
	}
	{
		float t1 = (lo.y - o.y) * inv.y, t2 = (hi.y - o.y) * inv.y;
		t_min = zy ? t_min : smax(t_min, smin(t1, t2));
		t_max = zy ? t_max : smin(t_max, smax(t1, t2));
	}
	{
		float t1 = (lo.z - o.z) * inv.z, t2 = (hi.z - o.z) * inv.z;
		t_min = zz ? t_min : smax(t_min, smin(t1, t2));
		t_max = zz ? t_max : smin(t_max, smax(t1, t2));
	}
	// a ray parallel to a slab and outside it misses (:113, :128, :139)
	const bool outside = (zx && (o.x < lo.x || o.x > hi.x)) || (zy && (o.y < lo.y || o.y > hi.y)) || (zz && (o.z < lo.z || o.z > hi.z));
	const bool ok = !outside && !(t_max < t_min) && (t_min > 0.0f || t_max > 0.0f);
	t = (t_min > 0.0f ? t_min : t_max) - kEps;
	return ok;
}
// The same test for a "clean" ray: origin, direction and reciprocals all finite, no direction
// component zero, the box free of NaN.  Then no t1 / t2 can be NaN (finite times finite, or an
// infinite bound times a finite non-zero reciprocal) and no axis takes the parallel branch, so
// the chain of std::max(t_min, std::min(t1, t2)) is a plain maximum of three minima, which the
// hardware's min / max / max3 form in 8 instructions instead of 24 compares and selects; they
// return the same numbers (a zero may come out with the other sign, which no later step sees:
// t_min is only compared with t_max and with 0 and has EPSILON subtracted).
__device__ __forceinline__ bool box_slab_clean(f3 lo, f3 hi, f3 o, f3 inv, float& t) {
	const float t1x = (lo.x - o.x) * inv.x, t2x = (hi.x - o.x) * inv.x;
	const float t1y = (lo.y - o.y) * inv.y, t2y = (hi.y - o.y) * inv.y;
	const float t1z = (lo.z - o.z) * inv.z, t2z = (hi.z - o.z) * inv.z;
	const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)), __builtin_fminf(t1z, t2z));
	const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)), __builtin_fmaxf(t1z, t2z));
	t = (t_min > 0.0f ? t_min : t_max) - kEps;
	return !(t_max < t_min) && (t_min > 0.0f || t_max > 0.0f);
}
// whether a ray is clean in that sense (per lane)
__device__ __forceinline__ bool ray_is_clean(f3 o, f3 d, f3 inv) {
	const float big = 3.0e38f;  // |x| < big: finite (and not NaN)
	// (a zero component of d has an infinite reciprocal, so it needs no test of its own)
	return fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big && fabsf(d.x) < big && fabsf(d.y) < big && fabsf(d.z) < big &&
	       fabsf(inv.x) < big && fabsf(inv.y) < big && fabsf(inv.z) < big;
}
__device__ __forceinline__ bool box_slab(f3 lo, f3 hi, f3 o, f3 d, float& t) {
	return box_slab_inv(lo, hi, o, d, mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), t);
}

// Box::surface_normal, framework/box.cpp:167-182
__device__ __forceinline__ f3 box_face_normal(f3 lo, f3 hi, f3 p) {
	if (p.x <= lo.x + kEps * 2) return {-1, 0, 0};
	else if (p.y <= lo.y + kEps * 2) return {0, -1, 0};
	else if (p.z <= lo.z + kEps * 2) return {0, 0, -1};
	else if (p.x >= hi.x - kEps * 2) return {1, 0, 0};
	else if (p.y >= hi.y - kEps * 2) return {0, 1, 0};
	else if (p.z >= hi.z - kEps * 2) return {0, 0, 1};
	return {0, 0, 0};
}

// Sphere::intersect up to `t` (framework/sphere.cpp:35-46 + glm gtx/intersect.inl:152-170);
// o is the ray origin in the sphere's local space, dn = normalize(d_local), len = length(d_local).
__device__ __forceinline__ bool sphere_t_pre(f3 c, float radius, f3 o, f3 dn, float len, float& t) {
	const float feps = 1.1920928955078125e-07f;  // std::numeric_limits<float>::epsilon()
	float r2 = radius * radius;
	f3 diff = c - o;
	float t0 = dot(diff, dn);
	float dsq = dot(diff, diff) - t0 * t0;
	const bool miss = dsq > r2;
	float t1 = __builtin_sqrtf(r2 - dsq);
	float tt = t0 > t1 + feps ? t0 - t1 : t0 + t1;
	const bool ok = !miss && (tt > feps);
	tt /= len;
	tt -= kEps;
	t = tt;
	return ok;
}
__device__ __forceinline__ bool sphere_t(f3 c, float radius, f3 o, f3 d, float& t) {
	return sphere_t_pre(c, radius, o, normalize(d), length(d), t);
}

// Triangle::intersect up to `t` (framework/triangle.cpp:56-85) with e1 = v1 - v0 and
// e2 = v2 - v0 formed beforehand (same single fp32 subtraction the reference does per call).
__device__ __forceinline__ bool tri_t(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float& t) {
	f3 p = cross(d, e2);
	float det = dot(e1, p);
	const bool parallel = det < kEps && det > -kEps;
	float inv_det = 1.0f / det;
	f3 tv = o - v0;
	float u = dot(tv, p) * inv_det;
	const bool bad_u = u < 0.0f || u > 1.0f;
	f3 q = cross(tv, e1);
	float v = dot(d, q) * inv_det;
	const bool bad_v = v < 0.0f || u + v > 1.0f;
	float tt = dot(e2, q) * inv_det;
	const bool bad_t = tt < kEps;
	t = tt - kEps;
	return !(parallel || bad_u || bad_v || bad_t);
}

// first two rejections of tri_t (same operations, so the compiler shares them with tri_t):
// false when the ray is parallel to the triangle or u is outside [0, 1]
__device__ __forceinline__ bool tri_u_ok(f3 v0, f3 e1, f3 e2, f3 o, f3 d) {
	f3 p = cross(d, e2);
	float det = dot(e1, p);
	const bool parallel = det < kEps && det > -kEps;
	float inv_det = 1.0f / det;
	f3 tv = o - v0;
	float u = dot(tv, p) * inv_det;
	return !(parallel || u < 0.0f || u > 1.0f);
}

// Renderer::schlick_reflection_ratio, framework/renderer.cpp:252-276
__device__ __forceinline__ float schlick(f3 rd, f3 n, float ior) {
	float n1 = 1.0f, n2 = ior;
	float cos_in = -dot(n, rd);
	if (cos_in < 0.0f) {
		float tmp = n1;
		n1 = n2;
		n2 = tmp;
	}
	if (n1 > n2) {
		float eta = n1 / n2;
		float s2 = eta * eta * (1.0f - cos_in * cos_in);
		if (s2 >= 1.0f) return 1.0f;
		cos_in = __builtin_sqrtf(1.0f - s2);
	}
	float r0 = (1.0f - ior) / (1.0f + ior);
	r0 *= r0;
	float f = 1.0f - cos_in;
	return r0 + (1.0f - r0) * f * f * f * f * f;
}

}  // namespace ptxd
