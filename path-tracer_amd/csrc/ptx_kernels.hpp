// Device side of the gfx950 render path: scene tables as the kernels see them, traversal,
// shading, queue plumbing and the kernels themselves.  Included by ptx.hip only (the host
// side there fills the structures declared here and launches the kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptx.h"
#include "ptx_bvh.hpp"
#include "ptx_device.hpp"

using namespace ptxd;

namespace {

constexpr int kBlock = 256;  // 4 wave64 per workgroup
// Workgroup of the kernels that walk (closest hit, shadow rays, the fused kernel, the parity tap).  Their waves share
// nothing - every wave has its own queue segment, stack and ring - so one wave is a workgroup of its own: a
// workgroup's registers and LDS are released only when its LAST wave ends, and waves whose segments cost unequal
// time would otherwise hold three finished waves' slots empty until the slowest is through.
#ifndef PTX_WALK_BLOCK
#define PTX_WALK_BLOCK 64
#endif
constexpr int kWalkBlock = PTX_WALK_BLOCK;
#ifndef PTX_SHADE_BLOCK
#define PTX_SHADE_BLOCK 256
#endif
// workgroup of the shading kernel: its waves share the LDS copies of the scene's small tables (measured, ms per image on
// the mesh scene, primary / deeper depths: 256 threads 30.0 / 76.1, 128 threads 33.7 / 76.7, 64 threads 38.9 / 80.2)
constexpr int kShadeBlock = PTX_SHADE_BLOCK;
#ifndef PTX_MIN_WAVES
#define PTX_MIN_WAVES 2  // waves per SIMD the bounce kernel's register budget is sized for
#endif
// Traversal stack: entries per lane, in LDS.  A BVH whose deepest leaf lies d levels below the root needs d; the deepest
// the builder makes is kBvhMaxDepth = 30.  The kernels that walk take the stack as dynamic LDS sized for the deepest
// tree of the scene at hand (walk_lds_bytes), because the stack is what limits the waves a compute unit holds, and these
// kernels run the faster the more waves there are to fill each other's waits (measured on the mesh scene, whose trees
// need 20 entries; ms per image, LDS per wave -> waves that fit 160 KiB: closest hit of depth 0 13.0 KiB -> 12 waves 45.5,
// 9.5 KiB -> 16 waves 37.0, 7.0 KiB -> 22 waves 32.3, 5.0 KiB -> its registers' 28 waves 28.5; of the deeper depths
// 8.0 KiB -> 20 waves 70.7, 7.0 KiB -> 22 waves 65.7, 6.5 KiB -> 24 waves 62.3; shadow rays 10 KiB 64.1, 8 KiB 60.1, 5 KiB
// 52.4; the image 272.9 with 30 entries everywhere, 250.3 as it is now).
constexpr int kStack = 30;
static_assert(kStack >= ptx::kBvhMaxDepth, "stack must cover the deepest BVH leaf");
constexpr int kMaxDepthSlots = 16;             // per-depth statistics (deeper levels share the last slot)
constexpr unsigned kMaxBounces = 63;
// A queue is cut into segments, one per wave of a launch (see the note on queues at the kernels).
#ifndef PTX_MAX_SEG
#define PTX_MAX_SEG 32768
#endif
// most segments = waves of a grid: five to eight rounds of the waves an MI355X keeps resident (six / four per SIMD in the
// walking / shading kernels).  Measured on the mesh scene (1920x1080, 1024 spp, 32 GiB of queues), ms per image: 16 384
// segments 264.3, 32 768 segments 250.8 (a launch ends with its last waves, and shorter waves make a shorter end),
// 65 536 segments 254.3 (with four waves per SIMD in the walking kernels it was 290.3 / 285.6 / 306).
constexpr unsigned kMaxSeg = PTX_MAX_SEG;
constexpr size_t count_bytes(unsigned max_seg) { return ((size_t) 3 * (kMaxBounces + 2) + 1) * max_seg * sizeof(unsigned); }  // per depth: ray, shadow-ray and hit counts of every segment; one more row: the order the segments of the depths > 0 are taken in

// ---------------------------------------------------------------------------
// device scene
// ---------------------------------------------------------------------------
// pointer to read-only scene data as the DEVICE pass sees it (constant address space, which
// lets uniform reads become scalar loads); the host pass only stores the address
#if defined(__HIP_DEVICE_COMPILE__)
#define PTX_KPTR(T) const __attribute__((address_space(4))) T*
#else
#define PTX_KPTR(T) const T*
#endif

constexpr int kPlain = 1 << 8;  // the shape's own transform is a pure translation: local = x + t, bit for bit

struct DevRootPrim {  // a leaf of frame 0, tested by every ray: 64 B, fetched with one wave-uniform scalar load
	float tinv[3];       // translation column of world_transform_inv_
	int32_t kind_flags;  // PtxPrimKind | kPlain
	float p[9];          // sphere: c, r | box: lo, hi | triangle: v0, e1, e2
	int32_t id;
	int32_t xform;       // index into DevScene::xforms (= the leaf's own slot: one entry per leaf)
	int32_t frame;       // 0: a leaf of the root; else the frame the leaf lies in (records behind the n_root_prims leaves of the root)
};
struct DevXform {  // full matrices of a leaf whose transform rotates or scales
	float winv[16];
	float w[16];
};
struct DevRootShade {  // what shading needs of a frame-0 leaf, per root slot
	float tw[3];  // translation column of world_transform_ (kept for inspection; shading uses the full matrices)
	int32_t material;
	float n[3];   // triangle: the normal a hit reports, finished at upload
	int32_t levels_in;
};
struct DevFrame {  // a mesh Composite with its own transform and bounds: 64-B head + matrices
	float bmin[3];
	int32_t bvh_root;   // child reference, node indices relative to node_base
	float bmax[3];
	int32_t node_base;  // first float4 of this frame's nodes
	float tinv[3];      // translation column of world_transform_inv_
	int32_t tri_base;   // first float4 of this frame's triangle records
	int32_t levels_up;
	int32_t has_prims;
	int32_t plain;      // transform is a pure translation
	int32_t parent;     // the frame this one lies in (0: the root)
};
struct Mat16 {
	float m[16];
};
struct DevScene {
	const DevRootPrim* root_prims;
	const DevRootShade* root_shade;
	const DevXform* xforms;
	const DevFrame* frames;
	const Mat16* frame_winv;
	const Mat16* frame_w;
	const float4* nodes;      // 4 float4 per node
	const float4* tris;       // 3 float4 per triangle: (v0, e1.x) (e1.yz, e2.xy) (e2.z, id, -, -)
	const float4* prim_info;  // per prim id: the normal a hit on a triangle reports, finished at upload (w: bits(material | levels_in << 16))
	const float4* materials;  // 4 float4 per PRIM id (its material, so that a hit needs no second, dependent fetch):
	                          // (ka, kd.x) (kd.yz, ks.xy) (ks.z, m, glossy, opacity) (ior,-,-,-)
	const float4* mat_table;  // the same 4 float4 per MATERIAL id (the shading kernel keeps a copy in LDS)
	const float4* lights;     // 2 float4 per light: (pos, -) (intensity, -)
	// the same tables once more, for the wave-uniform reads (kload)
	PTX_KPTR(DevRootPrim) root_prims_k;
	PTX_KPTR(DevXform) xforms_k;
	PTX_KPTR(DevFrame) frames_k;
	PTX_KPTR(Mat16) frame_winv_k;
	PTX_KPTR(float4) lights_k;
	PTX_KPTR(float4) root_bound_k;  // per frame-0 leaf: padded bounding box in frame-0 space, 2 x float4 (lo, hi)
	int32_t n_root_prims, n_frames, n_lights, n_prims, n_materials;
	int32_t has_plain_sphere;
	int32_t bounds_clean;    // no NaN in the root's or any mesh's bounds, no NaN / infinity in a frame-0 box or its translation
	int32_t root_bvh;        // != 0: the leaves of frame 0 are many and sit under a BVH of their own (root_walk)
	int32_t root_bvh_root;   // its root reference
	int32_t root_node_base;  // first float4 of its nodes in `nodes`
	int32_t nested;          // != 0: some frame lies inside another frame than the root, or holds leaves other than plain triangles (enter_nested)
	const int2* frame_leaves;  // per frame: (first, count) of its leaves that are not plain triangles, in root_prims / root_shade / xforms behind the root's
	// Octants of a split root (more than 64 children, composite.cpp:125-160; n_guards == 0 otherwise).  The
	// reference reaches a child of the root only through an octant whose bounds, and those of the octants above
	// it, the ray enters; which occurrence that is also decides ties and how often the normal is re-normalised.
	int32_t n_guards;
	const float4* guards;     // 2 float4 per octant: (bounds min, bits(octant above or -1)) (bounds max, bits(levels - 1))
	const int2* occ;          // occurrences of the root's children, grouped per child in visiting order: (octant or -1, tie-break key)
	const int2* root_occ;     // per frame-0 leaf slot: (first occurrence, count)
	const int4* frame_occ;    // per frame: (first occurrence, count, lowest prim id of the frame, -)
	// What the reference answers a ray whose direction is NaN in every component (the child rays of a hit whose
	// normal is NaN, see nan_ray_hit): the first child of the root in visiting order that is not a sphere.
	int32_t nan_id, nan_frame, nan_slot, nan_extra;  // nan_id < 0: the scene holds nothing but spheres
	float nan_t;
	float ambient[3];
	float root_min[3], root_max[3];
	// world_transform_inv_ of the root and of an untransformed leaf: the COMPUTED inverse of the identity
	// (shape.cpp:3-7), whose zero entries carry signs (see enter_identity)
	float root_winv[16], leaf_winv[16];
};

struct DevCamera {  // Renderer::render's camera basis, framework/renderer.cpp:27-37
	float m[16];    // trans_mat columns u, v, -dir, pos
	float dist;     // img_plane_dist
	float aa_unit;  // 1.0f / aa_steps
	float half_w, half_h;
	uint32_t width, height, aa_steps;
	double inv_width, inv_aa;  // 1.0 / width, 1.0 / aa_steps (udivmod)
	uint32_t jitter_seed;      // 0 = regular grid
	uint32_t pad;
};

struct RayQueue {  // SoA planes: each store/load is 16 (8) B per lane, contiguous across the wave; 40 B per ray
	float4* a;     // (o.x, o.y, o.z, d.x)
	float4* b;     // (d.y, d.z, w.r, w.g)
	float2* c;     // (w.b, bits(pixel))
};

struct Hit {
	float t;
	int32_t id;     // prim id (index into PtxSceneDesc::prims), -1 = miss
	int32_t frame;  // 0 = root leaf
	int32_t slot;   // root leaf: index into root_prims
	int32_t extra;  // split root only: re-normalisations of the normal beyond the one every root child gets (octant levels)
};

constexpr int kTestKinds = 5;  // slab tests, triangle tests, sphere tests, box tests, ray transforms
struct Counters {  // cumulative over one ptx_render_accumulate (filled by k_stats_reduce)
	unsigned long long trace_rays, shadow_rays, shaded_hits, overflow;
	unsigned long long remainder;  // waves that found more records in their segment than a launch could take (see flag_remainder)
	unsigned long long per_depth[kMaxDepthSlots];
	unsigned long long per_depth_shadow[kMaxDepthSlots];
	unsigned long long per_depth_hits[kMaxDepthSlots];
	unsigned long long tests[2][kMaxDepthSlots][kTestKinds];  // [closest-hit kernel | shadow kernel], only with test counting on
};
constexpr size_t kCounterBytes = 4096;
static_assert(sizeof(Counters) <= kCounterBytes, "counter block");
// Work counters are spread over kStatWaves slots of kStatWords words:
// stat[(depth_slot * kStatWaves + wave % kStatWaves) * kStatWords + word], every wave adding to its
// slot once, at its end, with no-return atomics.  Thousands of waves adding to ONE word
// serialise at the memory side (about 10 ns per add), which costs more than the rays they
// count; spread over thousands of words the adds proceed in parallel, and because every add
// is an atomic it does not matter how many waves a grid has.
//   word 0 closest-hit queries   1 shadow rays emitted   2 shaded hits
//   3..7   tests of the closest-hit kernel (kTestKinds)   8..12 tests of the shadow kernel
constexpr unsigned kStatWaves = 4096;
constexpr unsigned kStatWords = 16;
constexpr size_t kStatBytes = (size_t) kMaxDepthSlots * kStatWaves * kStatWords * sizeof(unsigned long long);

// Test counting is a compile-time policy: the kernels measured for speed carry NoCount (every
// call below vanishes), their counting twins carry TestCount and run when the caller asks for
// the numbers.  Counted per lane that executes the test.
struct NoCount {
	__device__ __forceinline__ void slab(unsigned = 1) {}
	__device__ __forceinline__ void tri() {}
	__device__ __forceinline__ void sphere() {}
	__device__ __forceinline__ void box() {}
	__device__ __forceinline__ void xform() {}
	__device__ __forceinline__ void needed() {}
	__device__ __forceinline__ unsigned get(int) const { return 0; }
};
struct TestCount {
	unsigned n[kTestKinds] = {0, 0, 0, 0, 0};
	__device__ __forceinline__ void slab(unsigned k = 1) { n[0] += k; }
	__device__ __forceinline__ void tri() { ++n[1]; }
	__device__ __forceinline__ void sphere() { ++n[2]; }
	__device__ __forceinline__ void box() { ++n[3]; }
#ifdef PTX_CENSUS_NEEDED
	// a build for one question - how much of the frame-0 leaf work is owed to the other rays of the wave? - in which
	// the last counter holds, per lane, the leaf tests the lane's OWN ray asked for (its own filter passed), to set
	// against the sphere / box / triangle counters, which count every lane of a wave that ran a test
	__device__ __forceinline__ void xform() {}
	__device__ __forceinline__ void needed() { ++n[4]; }
#else
	__device__ __forceinline__ void xform() { ++n[4]; }
	__device__ __forceinline__ void needed() {}
#endif
	__device__ __forceinline__ unsigned get(int k) const { return n[k]; }
};

__device__ __forceinline__ f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// Wave-uniform read of a small read-only record: the scene tables are never written while a
// kernel runs, so the kernel reads them through pointers in the constant address space,
// i.e. with scalar loads into SGPRs (one s_load_dwordx16 for a 64-B record) instead of 64
// identical vector loads.  The address MUST be the same in every lane.
template <class T>
__device__ __forceinline__ T kload(PTX_KPTR(T) p) {
	T v;
	__builtin_memcpy(&v, p, sizeof(T));
	return v;
}

// ---------------------------------------------------------------------------
// BVH walk.  The node test is a conservative filter (boxes are padded at build time and
// the far limit carries a margin above the reference's epsilon), so which leaves get
// tested never changes the answer: every triangle test is the reference's arithmetic and
// the winner is min t, ties to the lowest prim id.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bvh_slab2(float4 q0, float4 q1, float4 q2, f3 inv, f3 noi, float limit, float& tn0, bool& h0,
                                          float& tn1, bool& h1) {
	float ax = __builtin_fmaf(q0.x, inv.x, noi.x), bx = __builtin_fmaf(q0.w, inv.x, noi.x);
	float ay = __builtin_fmaf(q0.y, inv.y, noi.y), by = __builtin_fmaf(q1.x, inv.y, noi.y);
	float az = __builtin_fmaf(q0.z, inv.z, noi.z), bz = __builtin_fmaf(q1.y, inv.z, noi.z);
	tn0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
	float tf0 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), limit));
	h0 = tn0 <= tf0;
	float cx = __builtin_fmaf(q1.z, inv.x, noi.x), dx = __builtin_fmaf(q2.y, inv.x, noi.x);
	float cy = __builtin_fmaf(q1.w, inv.y, noi.y), dy = __builtin_fmaf(q2.z, inv.y, noi.y);
	float cz = __builtin_fmaf(q2.x, inv.z, noi.z), dz = __builtin_fmaf(q2.w, inv.z, noi.z);
	tn1 = fmaxf(fmaxf(fminf(cx, dx), fminf(cy, dy)), fmaxf(fminf(cz, dz), 0.0f));
	float tf1 = fminf(fminf(fmaxf(cx, dx), fmaxf(cy, dy)), fminf(fmaxf(cz, dz), limit));
	h1 = tn1 <= tf1;
}

__device__ __forceinline__ float bvh_rcp(float d) {
	float r = __builtin_amdgcn_rcpf(d);
	return fabsf(r) > 1e18f ? copysignf(1e18f, r) : r;
}

// the same clamp applied to a reciprocal already at hand
__device__ __forceinline__ float finite_rcp(float r) { return fabsf(r) > 1e18f ? copysignf(1e18f, r) : r; }

// far limit for node culling given the best reported distance (reported = raw - kEps)
__device__ __forceinline__ float cull_limit(float best_t) { return best_t + (2.0f * kEps + 1e-5f * fabsf(best_t)); }

#ifdef PTX_POP_CULL
// A build for one question - what does culling at pop time give? - in which a stack entry of the closest-hit walk carries,
// beside the child reference (17 bits and a leaf flag: meshes up to 16 384 triangles), the distance at which the ray
// enters the child's box, rounded down to 14 bits; an entry that lies beyond the best hit found since is dropped when
// it is taken off the stack instead of being visited.  Measured on the mesh-bound view (buster_closeup.sdf, 1920x1080,
// 1024 spp): primary closest-hit kernel 87.0 -> 85.0 ms, deeper depths 153.7 -> 154.3 ms, the image +0.2 %; not kept.
__device__ __forceinline__ int pack_entry(int ref, float tn) {
	const unsigned leaf = ref < 0 ? 1u : 0u;
	const unsigned v = (unsigned) (ref < 0 ? ~ref : ref) & 0x1ffffu;
	return (int) ((leaf << 31) | (v << 14) | ((__float_as_uint(tn) >> 17) & 0x3fffu));
}
__device__ __forceinline__ int entry_ref(int e) {
	const int v = (int) (((unsigned) e >> 14) & 0x1ffffu);
	return e < 0 ? ~v : v;
}
__device__ __forceinline__ float entry_distance(int e) { return __uint_as_float(((unsigned) e & 0x3fffu) << 17); }
#endif
template <bool ANY, class TC>
__device__ __forceinline__ bool bvh_walk(const float4* __restrict__ nodes, const float4* __restrict__ tris, int bvh_root, int frame, f3 o, f3 d,
                                         int* __restrict__ stk, Hit& best, float range, TC& tc, int id_bias = 0, int extra = 0) {
	// Reciprocal direction for the node tests only.  A zero (or denormal-small) component
	// would give inf and then inf - inf = NaN inside the fused slab test, which fminf/fmaxf
	// would read as "no constraint on the wrong side"; a huge finite reciprocal keeps the
	// test ordered, and the padding of the node boxes covers its rounding.
	f3 inv = {bvh_rcp(d.x), bvh_rcp(d.y), bvh_rcp(d.z)};
	f3 noi = {-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)};
	// Two nested loops instead of one loop that handles "inner node or leaf": a lane descends
	// through inner nodes until it stands on a leaf and then waits; the triangle tests start
	// when every lane of the wave stands on a leaf (or has finished).  The wave thus runs the
	// long triangle code once per round for all its lanes, not once per step of whichever
	// lane happens to be at a leaf.
	constexpr int kDone = (int) 0x80000000;  // negative like a leaf reference, never a valid one
	int sp = 0;
	int cur = bvh_root;
	float limit = ANY ? cull_limit(range) : cull_limit(best.t);
	// the next node from the stack, or kDone
	auto pop = [&](int& top, float far_limit) {
#ifdef PTX_POP_CULL
		while (top > 0) {
			--top;
			const int e = stk[top * kWalkBlock];
			if (ANY) return e;
			if (!(entry_distance(e) > far_limit)) return entry_ref(e);
		}
		return kDone;
#else
		(void) far_limit;
		if (top > 0) {
			--top;
			return stk[top * kWalkBlock];
		}
		return kDone;
#endif
	};
	while (cur != kDone) {
		while (cur >= 0) {
			const float4* nd = nodes + (size_t) cur * 4;
			float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			float tn0, tn1;
			bool h0, h1;
			bvh_slab2(q0, q1, q2, inv, noi, limit, tn0, h0, tn1, h1);
			tc.slab(2);
			int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
			if (h0 && h1) {
				if (tn1 < tn0) {
					int tmp = c0;
					c0 = c1;
					c1 = tmp;
#ifdef PTX_POP_CULL
					tn1 = tn0;
#endif
				}
#ifdef PTX_POP_CULL
				// the far child with the distance at which the ray enters it (rounded down to 14 bits): by the time it
				// is taken off the stack a hit nearer than that may have been found, and the visit is saved
				stk[sp * kWalkBlock] = ANY ? c1 : pack_entry(c1, tn1);
#else
				stk[sp * kWalkBlock] = c1;
#endif
				++sp;
				cur = c0;
			} else if (h0) {
				cur = c0;
			} else if (h1) {
				cur = c1;
			} else {
				cur = pop(sp, limit);
			}
		}
		if (cur == kDone) break;
		{
			int v = ~cur;
			int first = v >> 3, cnt = (v & 7) + 1;
			for (int k = 0; k < cnt; ++k) {
				const float4* tr = tris + (size_t) (first + k) * 3;
				float4 r0 = tr[0], r1 = tr[1], r2 = tr[2];
				float t;
				tc.tri();
				// most tests end at the first barycentric coordinate: when they do so for the whole wave, skip the rest
				if (!__ballot(tri_u_ok(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), o, d))) continue;
				if (tri_t(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), o, d, t)) {
					if (ANY) {
						if (t <= range) return true;
					} else {
						int id = __float_as_int(r2.y) + id_bias;  // (split root: the key of this occurrence of the mesh, see first_reachable)
						if (t < best.t || (t == best.t && id < best.id)) {
							best.t = t;
							best.id = id;
							best.frame = frame;
							best.extra = extra;
							best.slot = first + k;
							limit = cull_limit(t);
						}
					}
				}
			}
		}
		cur = pop(sp, limit);
	}
	return false;
}

// Per-ray values every test in frame-0 space shares.  They are the same IEEE operations the
// reference performs inside each Box / Sphere test (1 / d.x ..., normalize(d), length(d)),
// formed once: a shape whose transform is a pure translation sees the ray's direction
// unchanged, so its reciprocals, unit vector and length are these very numbers.
struct RayPre {
	f3 inv;     // 1.0f / d
	f3 dn;      // normalize(d)      (only when the scene holds a translated-only sphere)
	float len;  // length(d)
	f3 finv;    // finite reciprocal of d and -o * finv: the conservative bounding-box filter only
	f3 fnoi;
	bool clean;  // every ray of the wave is clean (box_slab_clean) and so are the scene's bounds: the same for all lanes
};
// Box::intersect(Ray, float&) in frame-0 space with the ray's shared reciprocals
__device__ __forceinline__ bool box_test(f3 lo, f3 hi, f3 o, f3 d, const RayPre& pre, float& t) {
	if (pre.clean) return box_slab_clean(lo, hi, o, pre.inv, t);
	return box_slab_inv(lo, hi, o, d, pre.inv, t);
}
__device__ __forceinline__ RayPre ray_pre(const DevScene& S, f3 o, f3 d) {
	RayPre r;
	r.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	r.finv = mk(finite_rcp(r.inv.x), finite_rcp(r.inv.y), finite_rcp(r.inv.z));  // from the exact quotient: no second reciprocal
	r.fnoi = mk(-(o.x * r.finv.x), -(o.y * r.finv.y), -(o.z * r.finv.z));
	r.clean = S.bounds_clean && __ballot(!ray_is_clean(o, d, r.inv)) == 0;
	r.dn = mk(0, 0, 0);
	r.len = 0;
	if (S.has_plain_sphere) {
		r.dn = normalize(d);
		r.len = length(d);
	}
	return r;
}

// Conservative filter in front of a leaf's exact test: false only if the ray cannot reach the
// leaf's padded bounding box (slab test with fused multiply-adds on a finite reciprocal, like
// the BVH node test).  It never produces a value; whatever passes gets the reference's
// arithmetic.  16 instructions against 45 to 100 for an exact test, and a wave whose 64 rays
// all fail (the usual case for the rays of one pixel: of the six walls of a room they see
// one) skips the leaf altogether.  NaN compares false and so passes.
// `limit`: a leaf the ray enters beyond it cannot matter (cull_limit of the best distance so far, or
// of a shadow ray's range; the ray parameter is the same in every space, so one limit serves all leaves).
__device__ __forceinline__ bool may_hit_box(float4 lo, float4 hi, const RayPre& pre, float limit) {
	float ax = __builtin_fmaf(lo.x, pre.finv.x, pre.fnoi.x), bx = __builtin_fmaf(hi.x, pre.finv.x, pre.fnoi.x);
	float ay = __builtin_fmaf(lo.y, pre.finv.y, pre.fnoi.y), by = __builtin_fmaf(hi.y, pre.finv.y, pre.fnoi.y);
	float az = __builtin_fmaf(lo.z, pre.finv.z, pre.fnoi.z), bz = __builtin_fmaf(hi.z, pre.finv.z, pre.fnoi.z);
	float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
	float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), limit));
	return !(tn > tf);
}

// one leaf of frame 0 against a frame-0-space ray: the reference's per-shape intersect up
// to the distance.  (1*x + 0*y) + (0*z + t*1) == x + t for finite operands, so a
// translated-only shape gets its local origin with one addition per component.
template <class TC>
__device__ __forceinline__ bool root_prim_t(const DevScene& S, const DevRootPrim& P, f3 o, f3 d, const RayPre& pre, float& t, TC& tc) {
	const int kind = P.kind_flags & 0xff;
	if (kind == PTX_SPHERE) tc.sphere();
	else if (kind == PTX_BOX) tc.box();
	else tc.tri();
	if (P.kind_flags & kPlain) {
		f3 lo = o + ld3(P.tinv);
		if (kind == PTX_SPHERE) return sphere_t_pre(ld3(P.p), P.p[3], lo, pre.dn, pre.len, t);
		if (kind == PTX_BOX) return box_test(ld3(P.p), ld3(P.p + 3), lo, d, pre, t);
		return tri_t(ld3(P.p), ld3(P.p + 3), ld3(P.p + 6), lo, d, t);
	}
	const Mat16 Wi = kload(reinterpret_cast<PTX_KPTR(Mat16)>(S.xforms_k + P.xform));
	f3 lo = xform_point(Wi.m, o), ld = xform_dir(Wi.m, d);
	tc.xform();
	if (kind == PTX_SPHERE) return sphere_t(ld3(P.p), P.p[3], lo, ld, t);
	if (kind == PTX_BOX) return box_slab(ld3(P.p), ld3(P.p + 3), lo, ld, t);
	return tri_t(ld3(P.p), ld3(P.p + 3), ld3(P.p + 6), lo, ld, t);
}

template <class TC>
__device__ __forceinline__ void frame_ray(const DevScene& S, const DevFrame& F, int f, f3 o, f3 d, f3& lo, f3& ld, TC& tc) {
	tc.xform();
	if (F.plain) {
		lo = o + ld3(F.tinv);
		ld = d;
	} else {
		const Mat16 Wi = kload(S.frame_winv_k + f);
		lo = xform_point(Wi.m, o);
		ld = xform_dir(Wi.m, d);
	}
}

// Split root: the first occurrence, in the reference's visiting order, of a child of the root (occurrences
// [first, first + count)) that the ray reaches, i.e. whose octant and every octant above it pass
// Composite::intersect's bounds test (composite.cpp:78-82; the root's own test is done before).  `key`: the
// occurrence's rank in visiting order - among equal distances the reference keeps the hit it met first (:93) -
// and `extra`: the octant levels above the child.  While a ray is traced, Hit::id holds keys (a mesh occurrence
// owns a run of them, one per triangle in prim order); hit_key_to_id puts the prim id back.
template <class TC>
__device__ __forceinline__ bool first_reachable(const DevScene& S, int first, int count, f3 o, f3 d, const RayPre& pre, int& key, int& extra, TC& tc) {
	for (int k = 0; k < count; ++k) {
		const int2 oc = S.occ[first + k];
		int g = oc.x, ex = 0;
		bool open = true;
		if (g >= 0) ex = __float_as_int(S.guards[2 * g + 1].w);
		while (g >= 0 && open) {
			const float4 lo = S.guards[2 * g], hi = S.guards[2 * g + 1];
			float unused;
			tc.slab();
			open = box_test(mk(lo.x, lo.y, lo.z), mk(hi.x, hi.y, hi.z), o, d, pre, unused);
			g = __float_as_int(lo.w);
		}
		if (open) {
			key = oc.y;
			extra = ex;
			return true;
		}
	}
	return false;
}
__device__ __forceinline__ void hit_key_to_id(const DevScene& S, Hit& h) {
	if (!S.n_guards || h.id < 0) return;
	h.id = h.frame == 0 ? S.root_prims[h.slot].id : __float_as_int(S.tris[(size_t) S.frames[h.frame].tri_base + (size_t) h.slot * 3 + 2].y);
}

// The same test with the leaf chosen per lane (ordinary loads instead of scalar ones): the
// leaves of a large frame 0 are reached through a BVH, and lanes arrive at different ones.
template <class TC>
__device__ __forceinline__ bool root_prim_t_lane(const DevScene& S, int slot, f3 o, f3 d, const RayPre& pre, float& t, int& id, TC& tc) {
	const DevRootPrim& P = S.root_prims[slot];
	const int kf = P.kind_flags, kind = kf & 0xff;
	id = P.id;
	if (kind == PTX_SPHERE) tc.sphere();
	else if (kind == PTX_BOX) tc.box();
	else tc.tri();
	if (kf & kPlain) {
		f3 lo = o + ld3(P.tinv);
		if (kind == PTX_SPHERE) return sphere_t_pre(ld3(P.p), P.p[3], lo, pre.dn, pre.len, t);
		if (kind == PTX_BOX) return box_test(ld3(P.p), ld3(P.p + 3), lo, d, pre, t);
		return tri_t(ld3(P.p), ld3(P.p + 3), ld3(P.p + 6), lo, d, t);
	}
	const float* Wi = S.xforms[slot].winv;
	f3 lo = xform_point(Wi, o), ld = xform_dir(Wi, d);
	tc.xform();
	if (kind == PTX_SPHERE) return sphere_t(ld3(P.p), P.p[3], lo, ld, t);
	if (kind == PTX_BOX) return box_slab(ld3(P.p), ld3(P.p + 3), lo, ld, t);
	return tri_t(ld3(P.p), ld3(P.p + 3), ld3(P.p + 6), lo, ld, t);
}

// bvh_walk over the leaves of frame 0 (built over their padded bounding boxes when there are
// more than a handful; the reference walks its root's children one by one,
// composite.cpp:75-103).  Same conservative node test, same two nested loops; a leaf holds up
// to four shapes of any kind, each tested with the reference's arithmetic.
template <bool ANY, class TC>
__device__ __forceinline__ bool root_walk(const DevScene& S, f3 o, f3 d, const RayPre& pre, int* __restrict__ stk, Hit& best, float range, TC& tc) {
	const float4* __restrict__ nodes = S.nodes + S.root_node_base;
	const f3 inv = pre.finv, noi = pre.fnoi;
	constexpr int kDone = (int) 0x80000000;
	int sp = 0;
	int cur = S.root_bvh_root;
	float limit = ANY ? cull_limit(range) : cull_limit(best.t);
	while (cur != kDone) {
		while (cur >= 0) {
			const float4* nd = nodes + (size_t) cur * 4;
			float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			float tn0, tn1;
			bool h0, h1;
			bvh_slab2(q0, q1, q2, inv, noi, limit, tn0, h0, tn1, h1);
			tc.slab(2);
			int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
			if (h0 && h1) {
				if (tn1 < tn0) {
					int tmp = c0;
					c0 = c1;
					c1 = tmp;
				}
				stk[sp * kWalkBlock] = c1;
				++sp;
				cur = c0;
			} else if (h0) {
				cur = c0;
			} else if (h1) {
				cur = c1;
			} else if (sp > 0) {
				--sp;
				cur = stk[sp * kWalkBlock];
			} else {
				cur = kDone;
			}
		}
		if (cur == kDone) break;
		{
			int v = ~cur;
			int first = v >> 3, cnt = (v & 7) + 1;
			for (int k = 0; k < cnt; ++k) {
				float t;
				int id;
				if (root_prim_t_lane(S, first + k, o, d, pre, t, id, tc)) {
					int extra = 0;
					if (S.n_guards) {  // split root: only through an octant the ray enters
						if (ANY ? !(t <= range) : !(t <= best.t)) continue;
						const int2 ro = S.root_occ[first + k];
						if (!first_reachable(S, ro.x, ro.y, o, d, pre, id, extra, tc)) continue;
					}
					if (ANY) {
						if (t <= range) return true;
					} else if (t < best.t || (t == best.t && id < best.id)) {
						best.t = t;
						best.id = id;
						best.frame = 0;
						best.extra = extra;
						best.slot = first + k;
						limit = cull_limit(t);
					}
				}
			}
		}
		if (sp > 0) {
			--sp;
			cur = stk[sp * kWalkBlock];
		} else {
			cur = kDone;
		}
	}
	return false;
}

// A frame inside another frame - a transformed Composite below a transformed Composite, e.g. a sub-object of a loaded
// mesh that a caller moved (composite.cpp:75-103 recurses, one bounds test and one change of coordinates per level).
// The ray goes down the chain of frames from the root's child to `f`: at every level the reference's bounds test in
// the space the ray has there, then the level's own inverse.  false: the ray does not get as far as the inside of `f`.
// (`f` is the same in every lane; the chain is short, so the frame above is looked up again rather than kept)
constexpr int kMaxNest = 32;
template <class TC>
__device__ __forceinline__ bool enter_nested(const DevScene& S, int f, f3 o, f3 d, f3& lo, f3& ld, TC& tc) {
	int depth = 0;
	for (int a = f; a > 0; a = kload(S.frames_k + a).parent) ++depth;
	lo = o;
	ld = d;
	bool open = true;
	for (int k = depth - 1; k >= 0; --k) {
		int a = f;
		for (int j = 0; j < k; ++j) a = kload(S.frames_k + a).parent;
		const DevFrame F = kload(S.frames_k + a);
		float unused;
		tc.slab();
		open = open && box_slab(ld3(F.bmin), ld3(F.bmax), lo, ld, unused);
		if (!__ballot(open)) return false;
		f3 no, nd;
		frame_ray(S, F, a, lo, ld, no, nd, tc);
		lo = no;
		ld = nd;
	}
	return open;
}

// A sphere, a box or a moved triangle that lies directly in a frame other than the root (a group of shapes a caller
// built and moved; a face of a moved mesh part moved once more): the reference's per-shape intersect on the ray as
// it is inside that frame, through the leaf's own inverse (shape.cpp:50-52; sphere.cpp:35-46, box.cpp:89-165,
// triangle.cpp:56-85).  Leaf `slot` of root_prims, chosen per frame (the same in every lane).
template <class TC>
__device__ __forceinline__ bool frame_leaf_t(const DevScene& S, int slot, f3 o, f3 d, float& t, int& id, TC& tc) {
	const DevRootPrim P = kload(S.root_prims_k + slot);
	const Mat16 Wi = kload(reinterpret_cast<PTX_KPTR(Mat16)>(S.xforms_k + slot));
	const int kind = P.kind_flags & 0xff;
	id = P.id;
	const f3 lo = xform_point(Wi.m, o), ld = xform_dir(Wi.m, d);
	tc.xform();
	if (kind == PTX_SPHERE) {
		tc.sphere();
		return sphere_t(ld3(P.p), P.p[3], lo, ld, t);
	}
	if (kind == PTX_BOX) {
		tc.box();
		return box_slab(ld3(P.p), ld3(P.p + 3), lo, ld, t);
	}
	tc.tri();
	return tri_t(ld3(P.p), ld3(P.p + 3), ld3(P.p + 6), lo, ld, t);
}

// scene.root->intersect(ray): Composite::intersect of the root, framework/composite.cpp:75-103.
// In parts, so that the closest-hit kernel can run the mesh walks of several steps together
// (see k_trace):
//   closest_root    root bounds + the leaves of frame 0; false when the ray misses the root bounds
//   frame_mask      bit f set: the ray enters the bounds of mesh frame f (frames 1..31; bit 0
//                   set: there are frames beyond 31, which closest_frames tests itself)
//   closest_frames  the mesh frames; with `known` a frame below 32 is walked iff its bit is set
// (RB: the kernel variant for scenes whose frame-0 leaves sit under a BVH; the variant without
// it still gives the same answers on such a scene, leaf by leaf)
template <bool RB, class TC>
__device__ __forceinline__ bool closest_root(const DevScene& S, f3 o, f3 d, const RayPre& pre, int* __restrict__ stk, Hit& best, TC& tc) {
	best.t = kInf;
	best.id = -1;
	best.frame = 0;
	best.slot = 0;
	best.extra = 0;
	float unused;
	tc.slab();
	if (!box_test(ld3(S.root_min), ld3(S.root_max), o, d, pre, unused)) return false;
	if (RB && S.root_bvh) {
		root_walk<false>(S, o, d, pre, stk, best, 0.0f, tc);
		return true;
	}
	for (int i = 0; i < S.n_root_prims; ++i) {
		tc.slab();
		const bool may = may_hit_box(kload(S.root_bound_k + 2 * i), kload(S.root_bound_k + 2 * i + 1), pre, cull_limit(best.t));
		if (may) tc.needed();
		if (!__ballot(may)) continue;
		const DevRootPrim P = kload(S.root_prims_k + i);
		float t;
		bool hit = root_prim_t(S, P, o, d, pre, t, tc);
		int key = P.id, extra = 0;
		if (RB && S.n_guards) {  // split root: only through an octant the ray enters
			if (hit && t <= best.t) {
				const int2 ro = S.root_occ[i];
				hit = first_reachable(S, ro.x, ro.y, o, d, pre, key, extra, tc);
			}
		}
		const bool better = hit && (t < best.t || (t == best.t && key < best.id));
		best.t = better ? t : best.t;
		best.id = better ? key : best.id;
		best.slot = better ? i : best.slot;
		if (RB) best.extra = better ? extra : best.extra;
	}
	return true;
}
template <bool RB, class TC>
__device__ __forceinline__ unsigned frame_mask(const DevScene& S, f3 o, f3 d, const RayPre& pre, TC& tc) {
	if (RB && S.nested) return 1u;  // frames inside frames: closest_frames tests every chain itself
	unsigned mask = S.n_frames > 32 ? 1u : 0u;
	float unused;
	const int nf = S.n_frames < 32 ? S.n_frames : 32;
	for (int f = 1; f < nf; ++f) {
		const DevFrame F = kload(S.frames_k + f);
		if (!F.has_prims) continue;
		tc.slab();
		if (box_test(ld3(F.bmin), ld3(F.bmax), o, d, pre, unused)) mask |= 1u << f;
	}
	return mask;
}
template <bool KNOWN, bool RB, class TC>
__device__ __forceinline__ void closest_frames(const DevScene& S, f3 o, f3 d, const RayPre& pre, unsigned known, int* __restrict__ stk, Hit& best, TC& tc) {
	float unused;
	if (RB && S.nested) {  // frames inside frames: every mesh through the whole chain of frames above it
		for (int f = 1; f < S.n_frames; ++f) {
			const DevFrame F = kload(S.frames_k + f);
			const int2 leaves = S.frame_leaves[f];
			if (!F.has_prims && !leaves.y) continue;
			int bias = 0, extra = 0;
			if (S.n_guards) {  // split root: only through an octant the ray enters, on the way to the root's child the chain starts at
				const int4 fo = S.frame_occ[f];
				int key;
				if (!first_reachable(S, fo.x, fo.y, o, d, pre, key, extra, tc)) continue;
				bias = key - fo.z;
			}
			f3 lo, ld;
			if (!enter_nested(S, f, o, d, lo, ld, tc)) continue;
			for (int j = leaves.x; j < leaves.x + leaves.y; ++j) {
				float t;
				int id;
				if (frame_leaf_t(S, j, lo, ld, t, id, tc) && (t < best.t || (t == best.t && id + bias < best.id))) {
					best.t = t;
					best.id = id + bias;
					best.frame = 0;  // (a leaf is named by its slot, like a leaf of the root)
					best.slot = j;
					best.extra = extra;
				}
			}
			if (F.has_prims) bvh_walk<false>(S.nodes + F.node_base, S.tris + F.tri_base, F.bvh_root, f, lo, ld, stk, best, 0.0f, tc, bias, extra);
		}
		return;
	}
	for (int f = 1; f < S.n_frames; ++f) {
		const DevFrame F = kload(S.frames_k + f);
		if (!F.has_prims) continue;
		if (KNOWN && f < 32) {
			if (!(known >> f & 1u)) continue;
		} else {
			tc.slab();
			if (!box_test(ld3(F.bmin), ld3(F.bmax), o, d, pre, unused)) continue;
		}
		int bias = 0, extra = 0;
		if (RB && S.n_guards) {  // split root: the mesh only through an octant the ray enters
			const int4 fo = S.frame_occ[f];
			int key;
			if (!first_reachable(S, fo.x, fo.y, o, d, pre, key, extra, tc)) continue;
			bias = key - fo.z;
		}
		f3 lo, ld;
		frame_ray(S, F, f, o, d, lo, ld, tc);
		bvh_walk<false>(S.nodes + F.node_base, S.tris + F.tri_base, F.bvh_root, f, lo, ld, stk, best, 0.0f, tc, bias, extra);
	}
}
// A ray whose direction is all NaN.  A box hit whose point matches none of the six faces reports the zero vector
// as its normal (box.cpp:167-182), Composite::intersect normalises it to NaN (composite.cpp:100), and the
// reflected / refracted rays of that hit have NaN directions.  The reference does not miss with such a ray: every
// comparison with NaN is false, so Box::intersect returns a hit at infinite distance (no slab constrains the
// interval, box.cpp:105-165), Triangle::intersect passes all its rejections and returns distance NaN
// (triangle.cpp:56-88), a sphere misses (`t > epsilon` is false), every bounds test passes, and
// Composite::intersect keeps the first child that hits because nothing compares less than it afterwards
// (composite.cpp:93).  The answer is therefore the same for every such ray: the first leaf in visiting order that
// is not a sphere, found at scene creation.  With a light in the scene the colour of that hit is NaN, and so is
// the pixel - also when the hit that spawned the ray lay in shadow and was itself finite.
__device__ __forceinline__ bool nan_direction(f3 d) { return d.x != d.x && d.y != d.y && d.z != d.z; }
__device__ __forceinline__ void nan_ray_hit(const DevScene& S, Hit& h) {
	h.id = S.nan_id;
	h.frame = S.nan_frame;
	h.slot = S.nan_slot;
	h.extra = S.nan_extra;
	h.t = S.nan_t;
}

template <bool RB = false>
__device__ __forceinline__ void trace_closest(const DevScene& S, f3 o, f3 d, int* __restrict__ stk, Hit& best) {
	if (nan_direction(d)) {
		nan_ray_hit(S, best);
		return;
	}
	const RayPre pre = ray_pre(S, o, d);
	NoCount tc;
	if (!closest_root<RB>(S, o, d, pre, stk, best, tc)) return;
	closest_frames<false, RB>(S, o, d, pre, 0u, stk, best, tc);
	if (RB) hit_key_to_id(S, best);
}

// find_light_block(...).does_intersect, framework/renderer.cpp:168-175: the closest hit
// lies within `range`  <=>  some leaf is hit at distance <= range.
template <bool RB, class TC>
__device__ __forceinline__ bool trace_any(const DevScene& S, f3 o, f3 d, float range, int* __restrict__ stk, TC& tc) {
	float t;
	RayPre pre = ray_pre(S, o, d);
	tc.slab();
	if (!box_test(ld3(S.root_min), ld3(S.root_max), o, d, pre, t)) return false;
	Hit dummy;
	if (RB && S.root_bvh) {
		if (root_walk<true>(S, o, d, pre, stk, dummy, range, tc)) return true;
	} else {
		const float limit = cull_limit(range);
		for (int i = 0; i < S.n_root_prims; ++i) {
			tc.slab();
			const bool may = may_hit_box(kload(S.root_bound_k + 2 * i), kload(S.root_bound_k + 2 * i + 1), pre, limit);
			if (may) tc.needed();
			if (!__ballot(may)) continue;
			const DevRootPrim P = kload(S.root_prims_k + i);
			if (root_prim_t(S, P, o, d, pre, t, tc) && t <= range) {
				if (RB && S.n_guards) {
					const int2 ro = S.root_occ[i];
					int key, extra;
					if (!first_reachable(S, ro.x, ro.y, o, d, pre, key, extra, tc)) continue;
				}
				return true;
			}
		}
	}
	for (int f = 1; f < S.n_frames; ++f) {
		const DevFrame F = kload(S.frames_k + f);
		if (RB && S.nested) {  // frames inside frames: through the whole chain of frames above the mesh
			const int2 leaves = S.frame_leaves[f];
			if (!F.has_prims && !leaves.y) continue;
			if (S.n_guards) {
				const int4 fo = S.frame_occ[f];
				int key, extra;
				if (!first_reachable(S, fo.x, fo.y, o, d, pre, key, extra, tc)) continue;
			}
			f3 lo, ld;
			if (!enter_nested(S, f, o, d, lo, ld, tc)) continue;
			bool found = false;
			for (int j = leaves.x; j < leaves.x + leaves.y; ++j) {
				float tl;
				int id;
				found = found || (frame_leaf_t(S, j, lo, ld, tl, id, tc) && tl <= range);
			}
			if (found) return true;
			if (F.has_prims && bvh_walk<true>(S.nodes + F.node_base, S.tris + F.tri_base, F.bvh_root, f, lo, ld, stk, dummy, range, tc)) return true;
			continue;
		}
		if (!F.has_prims) continue;
		tc.slab();
		if (!box_test(ld3(F.bmin), ld3(F.bmax), o, d, pre, t)) continue;
		if (RB && S.n_guards) {
			const int4 fo = S.frame_occ[f];
			int key, extra;
			if (!first_reachable(S, fo.x, fo.y, o, d, pre, key, extra, tc)) continue;
		}
		f3 lo, ld;
		frame_ray(S, F, f, o, d, lo, ld, tc);
		if (bvh_walk<true>(S.nodes + F.node_base, S.tris + F.tri_base, F.bvh_root, f, lo, ld, stk, dummy, range, tc)) return true;
	}
	return false;
}
template <bool RB = false>
__device__ __forceinline__ bool trace_any(const DevScene& S, f3 o, f3 d, float range, int* __restrict__ stk) {
	NoCount tc;
	return trace_any<RB>(S, o, d, range, stk, tc);
}

struct Surface {  // the HitPoint fields shading reads, framework/hitPoint.hpp:8-16
	f3 pos, n, rdir;
};

// Where shading reads the small per-leaf / per-mesh / per-material tables from: the device
// tables, or copies of them in LDS (k_shade).
struct ShadeTables {
	const DevRootPrim* prim;    // per frame-0 leaf
	const DevRootShade* shade;  // per frame-0 leaf
	const DevXform* xf;         // per frame-0 leaf
	const Mat16* fwinv;         // per frame
	const Mat16* fw;            // per frame
	const float4* mats;         // 4 float4 per material id
	const float* root_winv;     // DevScene::root_winv / leaf_winv (wave-uniform)
	const float* leaf_winv;
	const DevFrame* frames;     // only for frames inside frames (the device table; NULL in the copies that know none)
};
__device__ __forceinline__ ShadeTables device_tables(const DevScene& S) {
	return {S.root_prims, S.root_shade, S.xforms, S.frame_winv, S.frame_w, S.mat_table, S.root_winv, S.leaf_winv, S.nested ? S.frames : nullptr};
}

// Rebuilds position / normal / ray_direction of the winning leaf exactly as the leaf and
// the Composites above it would have produced them.  `info` = prim_info of the hit prim (read
// for mesh triangles only).
// A Composite hands the ray on to its children as transform_ray(ray, world_transform_inv_) (composite.cpp:83), and
// every leaf maps it by its own inverse again.  For the root, an octant, an OBJ sub-object or an untransformed
// triangle that matrix is the computed inverse of the identity: the product leaves every non-zero component as
// it is, but a zero component may come out with the other sign (the zero entries of the matrix carry signs).
// Applying it once is the same as applying it any number of times.  Distances and hit decisions never
// depend on the sign of a zero; the ray direction and position a hit REPORTS do, so the surface rebuild
// repeats the product - only when some lane of the wave holds a negative zero component at all.
// (a positive zero comes out of that product as it went in - its own term is +0, and +0 plus any zero is +0 - so
// only a NEGATIVE zero component can change)
__device__ __forceinline__ bool has_negative_zero(f3 v) {
	return __float_as_uint(v.x) == 0x80000000u || __float_as_uint(v.y) == 0x80000000u || __float_as_uint(v.z) == 0x80000000u;
}
// the same for a position handed up through a Composite with the identity as world_transform_ (composite.cpp:99):
// a negative zero becomes a positive one
__device__ __forceinline__ f3 up_position(f3 p) { return {p.x + 0.0f, p.y + 0.0f, p.z + 0.0f}; }

// A triangle of a mesh that lies in a frame inside another frame (enter_nested): every level of the reference's recursion
// spelled out.  Down: the root's inverse, then per frame of the chain the inverse of the identity once more if identity
// Composites (sub-objects, octants) lie between it and the frame above, and the frame's own inverse; the triangle
// reports its direction after its own inverse and the position on the ray it was handed (triangle.cpp:86).  Up: per
// frame its transform (composite.cpp:99), and the identity's for the Composites above it.
template <bool ORIGIN_DONE>
__device__ __forceinline__ Surface nested_surface(const ShadeTables& T, f3 o, f3 d, const Hit& h, float4 info) {
	Surface s;
	int depth = 0;
	for (int a = h.frame; a > 0; a = T.frames[a].parent) ++depth;
	f3 lo = ORIGIN_DONE ? o : xform_point(T.root_winv, o), ld = xform_dir(T.root_winv, d);
	for (int k = depth - 1; k >= 0; --k) {
		int a = h.frame;
		for (int j = 0; j < k; ++j) a = T.frames[a].parent;
		if (k != depth - 1 && T.frames[a].levels_up > 0) {
			lo = xform_point(T.leaf_winv, lo);
			ld = xform_dir(T.leaf_winv, ld);
		}
		lo = xform_point(T.fwinv[a].m, lo);
		ld = xform_dir(T.fwinv[a].m, ld);
	}
	const bool inside = ((__float_as_int(info.w) >> 16) & 0xff) != 0;
	const f3 ldz = xform_dir(T.leaf_winv, ld);
	f3 p = inside ? up_position(xform_point(T.leaf_winv, lo) + ldz * h.t) : lo + ld * h.t;
	for (int a = h.frame; a > 0; a = T.frames[a].parent) {
		p = xform_point(T.fw[a].m, p);
		if (T.frames[a].levels_up > 0) p = up_position(p);
	}
	s.pos = p;
	s.n = mk(info.x, info.y, info.z);  // constant per triangle: the whole chain applied at upload
	s.rdir = ldz;
	return s;
}

// The same for a sphere, a box or a moved triangle that lies directly in a frame other than the root (frame_leaf_t).
template <bool ORIGIN_DONE>
__device__ __forceinline__ Surface nested_leaf_surface(const ShadeTables& T, f3 o, f3 d, const Hit& h) {
	Surface s;
	const DevRootPrim& P = T.prim[h.slot];
	const DevRootShade& Q = T.shade[h.slot];
	const DevXform& X = T.xf[h.slot];
	const int frame = P.frame, kind = P.kind_flags & 0xff;
	int depth = 0;
	for (int a = frame; a > 0; a = T.frames[a].parent) ++depth;
	f3 fo = ORIGIN_DONE ? o : xform_point(T.root_winv, o), fd = xform_dir(T.root_winv, d);
	for (int k = depth - 1; k >= 0; --k) {
		int a = frame;
		for (int j = 0; j < k; ++j) a = T.frames[a].parent;
		if (k != depth - 1 && T.frames[a].levels_up > 0) {
			fo = xform_point(T.leaf_winv, fo);
			fd = xform_dir(T.leaf_winv, fd);
		}
		fo = xform_point(T.fwinv[a].m, fo);
		fd = xform_dir(T.fwinv[a].m, fd);
	}
	// the leaf, on the ray as it is inside its frame (as surface_of does it for a leaf of the root)
	const f3 lo = xform_point(X.winv, fo), ld = xform_dir(X.winv, fd);
	f3 p, n;
	if (kind == PTX_TRIANGLE) {
		p = fo + fd * h.t;
		s.rdir = ld;
		n = ld3(Q.n);  // constant per triangle: the whole chain applied at upload
	} else {
		if (kind == PTX_SPHERE) {
			p = xform_point(X.w, lo + ld * h.t);
			n = normalize(p - xform_point(X.w, ld3(P.p)));
		} else {
			n = normalize(xform_dir(X.w, box_face_normal(ld3(P.p), ld3(P.p + 3), lo + ld * h.t)));
			p = fo + fd * h.t;
		}
		s.rdir = fd;
	}
	for (int a = frame; a > 0; a = T.frames[a].parent) {
		p = xform_point(T.fw[a].m, p);
		if (kind != PTX_TRIANGLE) n = normalize(xform_dir(T.fw[a].m, n));
		const int up = T.frames[a].levels_up;
		if (up > 0) p = up_position(p);
		if (kind != PTX_TRIANGLE)
			for (int k = 0; k < up; ++k) n = up_one_level(n);
	}
	s.pos = p;
	s.n = n;
	return s;
}

// ORIGIN_DONE: the origin went through the root's inverse already (the camera position, at set-up)
// GENERAL: the variant that also knows frames inside frames (the kernels of ordinary scenes carry none of that code)
template <bool ORIGIN_DONE = false, bool GENERAL = false>
__device__ __forceinline__ Surface surface_of(const ShadeTables& T, f3 o, f3 d, const Hit& h, float4 info) {
	if (GENERAL && T.frames) {
		if (h.frame != 0) return nested_surface<ORIGIN_DONE>(T, o, d, h, info);
		if (T.prim[h.slot].frame != 0) return nested_leaf_surface<ORIGIN_DONE>(T, o, d, h);
	}
	Surface s;
	// (all the care for the signs of zeros sits behind wave-wide tests: a sum is a negative zero only if all its
	// terms are - an exact cancellation gives +0 - so a negative zero in a result needs one among the operands)
	// the ray as the root's children see it
	bool zeros = true;
	if (ORIGIN_DONE) {  // (primary rays, regenerated in registers: the product outright is cheaper than the test around it)
		d = xform_dir(T.root_winv, d);
	} else {
		const bool zero_o = __ballot(has_negative_zero(o)) != 0, zero_d = __ballot(has_negative_zero(d)) != 0;
		if (zero_o) o = xform_point(T.root_winv, o);
		if (zero_d) d = xform_dir(T.root_winv, d);
		zeros = zero_o || zero_d;
	}
	if (h.frame == 0) {
		const DevRootPrim& P = T.prim[h.slot];
		const DevRootShade& Q = T.shade[h.slot];
		const int kind = P.kind_flags & 0xff;
		// the shape's real matrices, also for translated-only shapes: the zeros of an inverted
		// translation carry signs, and the sign of a zero normal component follows them
		const DevXform& X = T.xf[h.slot];  // one per root leaf, in the same order
		f3 lo = xform_point(X.winv, o), ld = xform_dir(X.winv, d);
		if (kind == PTX_TRIANGLE) {  // framework/triangle.cpp:86
			s.pos = o + d * h.t;
			s.rdir = ld;
			s.n = ld3(Q.n);  // constant per triangle: transformed and re-normalised at upload
		} else {
			if (kind == PTX_SPHERE) {  // framework/sphere.cpp:47-48, :54-56
				s.pos = xform_point(X.w, lo + ld * h.t);
				s.n = normalize(s.pos - xform_point(X.w, ld3(P.p)));
				s.pos = up_position(s.pos);  // (after the normal: sphere.cpp forms it from its own position, the root maps the position afterwards)
			} else {  // framework/box.cpp:98-100
				f3 nl = box_face_normal(ld3(P.p), ld3(P.p + 3), lo + ld * h.t);
				s.n = normalize(xform_dir(X.w, nl));
				s.pos = o + d * h.t;
			}
			s.rdir = d;
			for (int k = 0; k < Q.levels_in; ++k) s.n = up_one_level(s.n);  // identity-transform Composites up to the root
		}
		if (zeros) s.pos = up_position(s.pos);  // through the root (and any octant above the leaf)
	} else {
		const float* Fwinv = T.fwinv[h.frame].m;  // h.frame differs between lanes: per-lane reads
		const float* Fw = T.fw[h.frame].m;
		f3 lo = xform_point(Fwinv, o), ld = xform_dir(Fwinv, d);
		f3 pl = lo + ld * h.t;
		const bool local_zeros = __ballot(has_negative_zero(lo) || has_negative_zero(ld) || has_negative_zero(pl)) != 0;
		if (local_zeros) {
			// Below the mesh's own Composite lie `inside` identity-transform Composites (sub-objects, octants) and
			// then the triangle.  The triangle reports the direction after its own inverse (triangle.cpp:86) and the
			// position on the ray it was handed: the mesh-local ray, mapped once more if a Composite lies between.
			const bool inside = ((__float_as_int(info.w) >> 16) & 0xff) != 0;
			ld = xform_dir(T.leaf_winv, ld);
			if (inside) pl = up_position(xform_point(T.leaf_winv, lo) + ld * h.t);
		}
		s.n = mk(info.x, info.y, info.z);  // constant per triangle: the whole chain of Composites applied at upload
		s.pos = xform_point(Fw, pl);
		if (local_zeros) s.pos = up_position(s.pos);  // through the root (and any octant above the mesh)
		s.rdir = ld;  // stays in mesh-local space: Composite never maps ray_direction back
	}
	return s;
}
template <bool GENERAL = false>
__device__ __forceinline__ Surface surface_of(const DevScene& S, f3 o, f3 d, const Hit& h) {
	float4 info = make_float4(0.f, 0.f, 0.f, 0.f);
	if (h.frame != 0) info = S.prim_info[h.id];
	Surface s = surface_of<false, GENERAL>(device_tables(S), o, d, h, info);
	for (int k = 0; k < h.extra; ++k) s.n = up_one_level(s.n);  // octants of a split root above the child (composite.cpp:100)
	return s;
}

struct Material {
	f3 ka, kd, ks;
	float m, glossy, opacity, ior;
};
__device__ __forceinline__ Material unpack_material(float4 a, float4 b, float4 c, float4 e) {
	return {{a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x}, c.y, c.z, c.w, e.x};
}
__device__ __forceinline__ Material load_material(const DevScene& S, int prim_id) {
	const float4* p = S.materials + (size_t) prim_id * 4;
	return unpack_material(p[0], p[1], p[2], p[3]);
}
// the material of a hit through the table by material id: a frame-0 leaf names its material in
// its shading record, a mesh triangle in the low half of prim_info.w
__device__ __forceinline__ Material material_of(const ShadeTables& T, const Hit& h, float4 info) {
	const int mid = h.frame == 0 ? T.shade[h.slot].material : (__float_as_int(info.w) & 0xffff);
	const float4* p = T.mats + (size_t) mid * 4;
	return unpack_material(p[0], p[1], p[2], p[3]);
}

// One light's term of Renderer::phong_color (framework/renderer.cpp:142-163) without the
// shadow test: direction and distance to the light, diffuse + specular (:177-192; pow is the
// double overload there).  Returns false when the term adds nothing (back-facing, or
// exactly zero), in which case the reference's shadow query cannot change the result.
struct LightTerm {
	f3 ldir, diffuse, spec;
	float dist;
	bool has_spec;
};
// (float) pow((double) x, (double) m) for a material exponent that is a whole number in
// [1, kIntPowMax]: square-and-multiply in double.  Every product is rounded once, so the
// result is within about m * 2^-54 of the true power, far inside what survives the final
// rounding to float (the general pow of the device library is as accurate but keeps some
// forty more registers live, which costs the shading kernels a wave per SIMD).  x >= 0, or
// NaN, which the products pass on like pow does.
constexpr float kIntPowMax = 4095.0f;
__device__ __forceinline__ float pow_int(float x, float m) {
	double b = (double) x, r = 1.0;
	unsigned e = (unsigned) m;
	while (__ballot(e != 0)) {
		if (e & 1u) r *= b;
		b *= b;
		e >>= 1;
	}
	return (float) r;
}

template <bool INTPOW>
__device__ __forceinline__ bool light_term(const DevScene& S, int l, const Surface& sf, f3 kd, f3 ks, float m, LightTerm& T) {
	const float4 lp = kload(S.lights_k + 2 * l), li = kload(S.lights_k + 2 * l + 1);
	f3 ldir = mk(lp.x, lp.y, lp.z) - sf.pos;
	T.dist = length(ldir);
	ldir = normalize(ldir);
	T.ldir = ldir;
	float c = dot(sf.n, ldir);
	f3 I = {li.x, li.y, li.z};
	T.diffuse = cmul(I, kd) * c;
	T.spec = mk(0, 0, 0);
	T.has_spec = (m != 0.0f);
	if (m != 0.0f && !(c < 0.0f)) {
		f3 r = 2 * dot(sf.n, ldir) * sf.n - ldir;
		float cs = dot(r, sf.rdir * -1.0f);
		if (!(cs < 0.0f)) {
			float f = INTPOW ? pow_int(cs, m) : (float) pow((double) cs, (double) m);
			T.spec = cmul(I, ks) * f;
		}
	}
	if (c < 0.0f) return false;
	return any_nonzero(T.diffuse) || any_nonzero(T.spec);
}

// Renderer::phong_color with the shadow walk inline (fused pipeline)
template <bool INTPOW>
__device__ __forceinline__ f3 phong(const DevScene& S, const Surface& sf, const Material& M, int* __restrict__ stk, unsigned& n_shadow) {
	f3 col = cmul(ld3(S.ambient), M.ka);
	for (int l = 0; l < S.n_lights; ++l) {
		LightTerm T;
		if (!light_term<INTPOW>(S, l, sf, M.kd, M.ks, M.m, T)) continue;
		++n_shadow;
		if (trace_any(S, sf.pos, T.ldir, T.dist, stk)) continue;
		col = col + T.diffuse;
		if (T.has_spec) col = col + T.spec;
	}
	return col;
}

struct Child {
	f3 o, d, w;
	bool live;
};

// The part of Renderer::shade after the Phong term (framework/renderer.cpp:117-134) with
// reflection (:194-201) and refraction (:203-225) turned into weighted child rays: the
// child's colour enters the parent linearly, so weight_child = weight * (factor the
// reference multiplies it by).  k1 is the factor the reference applies to the local colour
// after `phong * opacity` (1 when no case applies).
__device__ __forceinline__ void bounce_terms(const Surface& sf, const Material& M, f3 weight, bool last_depth, bool all_children, Child& refl,
                                             Child& refr, float& k1) {
	refl.live = false;
	refr.live = false;
	k1 = 1.0f;
	if (last_depth) return;

	float w_refl = 0.0f, w_refr = 0.0f;  // scalar factors on reflection()*ks and refraction()*kd
	bool want_refl = false, want_refr = false;
	if (M.glossy > 0.0f && M.opacity < 1.0f) {
		float R = schlick(sf.rdir, sf.n, M.ior);
		k1 = R * M.opacity;
		w_refl = R;
		w_refr = (1.0f - R) * (1.0f - M.opacity);
		want_refl = want_refr = true;
	} else if (M.glossy > 0.0f) {
		float R = schlick(sf.rdir, sf.n, M.ior);
		R = M.glossy + (1.0f - M.glossy) * R;
		k1 = 1.0f - R;
		w_refl = R;
		want_refl = true;
	} else if (M.opacity < 1.0f) {
		k1 = M.opacity;
		w_refr = 1.0f - M.opacity;
		want_refr = true;
	}
	f3 wr = {0, 0, 0};  // total weight on the reflection ray
	if (want_refl) wr = cmul(weight, M.ks) * w_refl;
	if (want_refr) {
		f3 rd = sf.rdir, n = sf.n;
		float eta = 1.0f / M.ior;
		float ci = -dot(n, rd);
		if (ci < 0.0f) {
			eta = 1.0f / eta;
			ci *= -1.0f;
			n = n * -1.0f;
		}
		float k = 1.0f - eta * eta * (1.0f - ci * ci);
		if (k < 0.0f) {
			// total internal reflection: refraction() returns reflection() (same ray, tinted by ks)
			f3 extra = cmul(weight, M.ks);
			if (M.glossy > 0.0f) extra = (extra * (1.0f - w_refl)) * (1.0f - M.opacity);
			else extra = extra * w_refr;
			wr = wr + extra;
			want_refl = true;
		} else {
			f3 wt = cmul(weight, M.kd);
			if (M.glossy > 0.0f) wt = (wt * (1.0f - w_refl)) * (1.0f - M.opacity);
			else wt = wt * w_refr;
			// A child whose weight is exactly zero (blue light on a mirror without blue in its ks, say) adds
			// nothing to the pixel and is not traced, unless the caller asks for every child: the reference traces
			// it and multiplies what comes back by that zero, which matters in one case only - a NaN from down there
			// (a box hit that matches no face, box.cpp:167-182) stays a NaN.
			refr.o = sf.pos - n * (2 * kEps);
			refr.d = rd * eta + n * (eta * ci - __builtin_sqrtf(k));
			refr.w = wt;
			refr.live = all_children || any_nonzero(wt);
		}
	}
	if (want_refl && (all_children || any_nonzero(wr))) {
		float ci = -dot(sf.n, sf.rdir);
		refl.o = sf.pos;
		refl.d = sf.rdir + (sf.n * ci * 2.0f);
		refl.w = wr;
		refl.live = true;
	}
}

// Renderer::shade for the fused pipeline: weight * (phong * opacity * k1), children out
template <bool INTPOW>
__device__ __forceinline__ f3 shade(const DevScene& S, const Surface& sf, int prim_id, f3 weight, bool last_depth, bool all_children, int* __restrict__ stk,
                                    Child& refl, Child& refr, unsigned& n_shadow) {
	Material M = load_material(S, prim_id);
	f3 c = phong<INTPOW>(S, sf, M, stk, n_shadow) * M.opacity;
	float k1;
	bounce_terms(sf, M, weight, last_depth, all_children, refl, refr, k1);
	c = c * k1;
	return cmul(weight, c);
}

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ unsigned long long lanes_below(unsigned lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// Adds `v` of the lanes with `has` into accum[pix*3..]: the wave is cut into runs of
// consecutive lanes with equal pixel (rays stay in pixel order through compaction), each
// run is summed with shuffles, and the run's first three lanes add its r, g and b to three
// consecutive words with ONE atomic instruction (float atomics execute at the memory side,
// one request per 64-byte line an instruction touches: three instructions of one lane each
// were three requests per run).  A run of one or two lanes adds the rest from its first lane.
// Sum over the 64 lanes of a wave, within the rows of 16 by data-parallel-primitive moves (plain
// vector instructions, no trip through the LDS crossbar), across the four rows by lane reads.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
	return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
	v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
	v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
	v = dpp_add<0x141>(v);  // row_half_mirror
	v = dpp_add<0x140>(v);  // row_mirror: every lane of a row holds the row's sum
	const int b = __float_as_int(v);
	return ((__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
	        __int_as_float(__builtin_amdgcn_readlane(b, 32))) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}

template <class AT>
__device__ __forceinline__ void accumulate_into(AT* __restrict__ accum, unsigned pix, f3 v, bool has) {
	const unsigned long long has_mask = __ballot(has);
	if (has_mask == 0) return;
	const unsigned lane = lane_id();
	if (!has) v = mk(0, 0, 0);
	{
		// the usual case at 64 or more samples per pixel: every lane that adds something adds to the
		// same pixel - one sum over the wave, one atomic instruction
		const unsigned first_pix = __builtin_amdgcn_readlane(pix, __builtin_ctzll(has_mask));
		if (__ballot(has && pix != first_pix) == 0) {
			const float sx = wave_sum(v.x), sy = wave_sum(v.y), sz = wave_sum(v.z);
			if (lane < 3) atomicAdd(accum + (size_t) first_pix * 3 + lane, (AT) (lane == 0 ? sx : lane == 1 ? sy : sz));
			return;
		}
	}
	const unsigned up = __shfl_up(pix, 1);
	const bool head = lane == 0 || up != pix;
	const unsigned long long heads = __ballot(head);
	// run heads above this lane (bit 0 = lane + 1): lane + off belongs to my run iff none of the
	// low `off` bits is set - the run bookkeeping needs no shuffles of its own
	const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		float ox = __shfl_down(v.x, off), oy = __shfl_down(v.y, off), oz = __shfl_down(v.z, off);
		if (lane + off < 64 && (above & ((1ull << off) - 1)) == 0) {
			v.x += ox;
			v.y += oy;
			v.z += oz;
		}
	}
	const unsigned run = above ? (unsigned) __builtin_ctzll(above) + 1 : 64 - lane;  // lanes from here to the end of my run
	const unsigned long long mine = has_mask >> lane;
	const bool live = head && (run >= 64 ? mine != 0 : (mine & ((1ull << run) - 1)) != 0);  // first lane of a run that has something to add
	const unsigned long long live_heads = __ballot(live);
	const float y1 = __shfl_up(v.y, 1), z2 = __shfl_up(v.z, 2);
	const bool second = !head && lane >= 1 && (live_heads >> (lane - 1) & 1);
	const bool third = !head && lane >= 2 && !(heads >> (lane - 1) & 1) && (live_heads >> (lane - 2) & 1);
	AT* p = accum + (size_t) pix * 3;
	if (live || second || third) atomicAdd(p + (live ? 0 : second ? 1 : 2), (AT) (live ? v.x : second ? y1 : z2));
	if (live && run < 2) atomicAdd(p + 1, (AT) v.y);
	if (live && run < 3) atomicAdd(p + 2, (AT) v.z);
}

// u / n and u % n for u < 2^31 with inv = 1.0 / n in double: the product is below the true
// quotient by less than one, so truncating it gives the quotient or one less, which the
// remainder shows.  (A handful of instructions; the compiler's 32-bit division takes about
// 25 and the 64-bit one well over a hundred, per ray.)
__device__ __forceinline__ void udivmod(unsigned u, unsigned n, double inv, unsigned& q, unsigned& r) {
	q = (unsigned) ((double) u * inv);
	r = u - q * n;
	if (r >= n) {
		++q;
		r -= n;
	}
}

__device__ __forceinline__ void make_primary(const DevCamera& cam, unsigned pixel, unsigned s, f3& o, f3& d) {
	// framework/renderer.cpp:66-67, :76-82
	unsigned x, y, ax, ay;
	udivmod(pixel, cam.width, cam.inv_width, y, x);
	udivmod(s, cam.aa_steps, cam.inv_aa, ax, ay);
	int aax = (int) ax, aay = (int) ay;
	f3 pp = {x + aax * cam.aa_unit - cam.half_w, y + aay * cam.aa_unit - cam.half_h, -cam.dist};
	if (cam.jitter_seed) {  // jittered sampling (not in the reference): a point inside the sample's cell
		float jx, jy;
		jitter_offsets(cam.jitter_seed, pixel, s, jx, jy);
		pp.x = x + (aax + jx) * cam.aa_unit - cam.half_w;
		pp.y = y + (aay + jy) * cam.aa_unit - cam.half_h;
	}
	f3 n = normalize(pp);
	d = xform_dir(cam.m, n);
	o = mk(cam.m[12], cam.m[13], cam.m[14]);
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// Queues.  Every queue (the rays of one depth, the shadow rays, the hit records) is cut into
// n_seg SEGMENTS of seg_cap slots, and segment w belongs to wave w of every launch: a grid is
// exactly n_seg waves.  Wave w of the closest-hit kernel reads ray segment w and writes hit
// segment w; wave w of the shading kernel reads hit segment w (and, through the slot a hit
// record names, ray segment w) and writes ray segment w of the next depth and shadow segment
// w; wave w of the shadow kernel reads shadow segment w.  So
//   * a wave appends by counting in a register and stores the count once, at its end: no
//     atomics on queue tails (one returning atomic per 64-record step and output queue was
//     what the shading kernel spent most of its time waiting for: doubling them doubled its
//     time), no prefix sums, no search: a record's place is (segment, index);
//   * segment w of an output holds at most (records taken from segment w of the input) x
//     (records one input can produce), whatever the wave does in between (parking rays,
//     walking them later): the host sizes `in_take` from that and nothing can overflow;
//   * the segments stay about equally full: the 64-sample steps of a batch are dealt to the
//     waves round-robin, so every wave gets an even sample of the image.
struct BounceArgs {
	DevScene S;
	DevCamera cam;
	RayQueue in, out;
	const unsigned* in_count;  // records in every segment of `in` (depth > 0), one word per segment
	unsigned* out_count;       // records in every segment of `out`: every wave of the launch writes its word
	unsigned seg_cap;          // slots per segment (the same for every queue)
	unsigned n_seg;            // segments = waves of the grid
	unsigned in_first;         // depth > 0: the launch takes records [in_first, in_first + in_take) of every segment of `in`
	unsigned in_take;          // (0 = all from in_first on)
	unsigned first_pix, first_s;  // depth 0: the batch starts at sample first_s (< ns) of this render's range in pixel first_pix
	double inv_ns;             // 1.0 / ns
	unsigned batch;            // depth 0: samples in the batch
	unsigned ns, s0;           // samples per pixel in this render's range, first sample
	unsigned depth;
	unsigned last_depth;       // depth == max_ray_bounces: shade locally only
	unsigned all_children;     // also trace child rays of exactly zero weight (PTX_TRACE_ZERO_WEIGHT)
	unsigned flag_remainder;   // depth > 0: count the waves whose segment holds more than the launch takes (nobody comes back for the rest)
	float* accum;              // the caller's accumulation buffer (added to directly only when accum64 is NULL)
	double* accum64;           // this render's own sums, folded into `accum` at its end
	Counters* counters;        // overflow / remainder flags
	unsigned long long* stat;  // work counters, spread over kStatWaves slots
	// staged pipeline only
	RayQueue shadow;           // a = (o.xyz, bits(pixel))  b = (contribution rgb, bits(light)); plane c unused (32 B per shadow ray)
	unsigned* shadow_count;
	unsigned stage_children;   // k_shade's staging area per wave: child rays (64, or 128 where a material spawns two) ...
	unsigned stage_lights;     // ... and lights whose shadow-ray records are staged (<= kStageLights)
	float4* hits;              // hit records: (t, bits(prim id), bits(where), bits(ray))
	                           // ray = the sample's index in the batch (depth 0) or the ray's slot in `in`
	unsigned* hit_count;
	const unsigned* order;     // wave i of the grid takes segment order[i] (fullest first, see k_order); NULL: segment i
};

// Where the sums of a render go.  A pixel of a 10 000-sample image receives some 10^5 additions; into a float that
// has grown to 10^4 each of them rounds at 10^-3, and together they show at the 10^-5 level of the finished pixel.
// A render therefore adds into a buffer of doubles of its own (BounceArgs::accum64, zero at its start), and the
// caller's float buffer receives one addition per pixel at its end (k_fold).  The sums inside a wave stay float
// (64 terms); the float buffer is added to directly only where the library may not hold memory of its own.
template <class ARGS>
__device__ __forceinline__ void accumulate(const ARGS& A, unsigned pix, f3 v, bool has) {
	if (A.accum64) accumulate_into(A.accum64, pix, v, has);
	else accumulate_into(A.accum, pix, v, has);
}

// Software pipelining of the queue kernels.  The memory counter of a wave (vmcnt) retires loads,
// stores and atomics strictly in issue order, and the compiler, which cannot count the stores of
// divergent branches, waits for ALL of them (vmcnt(0)) wherever a loaded value is first used.  A
// step that stores or adds its results at its end and then loads the next step's records
// therefore stands still for a full trip to memory (atomics: several microseconds under load)
// once or twice per step.  So every step issues ALL its memory traffic in one burst at its TOP -
// the stores / atomics of the step before (kept in registers or LDS until then) and the loads
// of the step after - and then computes; by the next top everything has long arrived, and the one
// wait there (pipeline_wait) is free.
__device__ __forceinline__ void pipeline_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // s_waitcnt vmcnt(0)

// Kernel arguments read in place.  BounceArgs is some 900 bytes, far more than the hundred scalar registers of a
// wave: taken by value, the compiler loads every member it will ever need once, at the top of the kernel, and then
// keeps what does not fit in the lanes of a vector register - a v_readlane per use inside the loop (the shading
// kernels executed about a hundred of them per 64-record step) and a dozen vector registers to hold them.  The
// queue kernels therefore read the arguments where the launch put them, through a pointer into the kernel-argument
// segment (constant address space: scalar loads that hit the scalar cache), and hide that pointer from the
// optimiser once per step, so that a member is loaded in the step that uses it and its register is free again
// afterwards.  (The by-value parameter stays in the signature: it is what puts the bytes there.)
typedef const __attribute__((address_space(4))) struct BounceArgs KArgs;
#define PTX_ARGS_IN_PLACE(ka) KArgs* ka = (KArgs*) __builtin_amdgcn_kernarg_segment_ptr()
#define PTX_ARGS_AGAIN(ka) asm volatile("" : "+s"(ka))
// this wave's number: the same in every lane, and known to the compiler to be (scalar registers, scalar loads)
template <int BLOCK = kBlock>
__device__ __forceinline__ unsigned wave_id() { return __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)); }

// The segment wave `w` of a grid works on.  The segments of a depth > 0 are unequally full - a wave's few pixels lie
// on glass or they do not - and a grid ends when its last wave ends: taken in the order of their numbers, a full
// segment that starts late keeps the chip waiting, nearly empty, for one wave.  k_order lists them fullest first.
__device__ __forceinline__ unsigned segment_of(const unsigned* order, unsigned w) {
	return order ? __builtin_amdgcn_readfirstlane(order[w >> 2]) * 4 + (w & 3u) : w;  // (the list holds runs of four segments)
}
#ifdef PTX_TIMELINE
// A build for one question - when do the waves of a launch start and end, and where? - in which every wave of the queue
// kernels leaves (start, end, records, hardware id) in a table the host dumps (scripts/timeline.py).
__device__ unsigned long long g_timeline[4 * kMaxSeg];
struct WaveTimer {
	unsigned long long t0;
	unsigned slot, seg;
	__device__ __forceinline__ WaveTimer(unsigned wave_slot, unsigned segment) : t0(wall_clock64()), slot(wave_slot), seg(segment) {}
	__device__ __forceinline__ void done(unsigned records) const {
		if (lane_id() == 0 && slot < kMaxSeg) {
			unsigned hw, xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			g_timeline[4 * slot] = t0;
			g_timeline[4 * slot + 1] = wall_clock64();
			g_timeline[4 * slot + 2] = records | ((unsigned long long) seg << 32);
			g_timeline[4 * slot + 3] = (unsigned long long) hw | ((unsigned long long) xcc << 32);
		}
	}
};
#else
struct WaveTimer {
	__device__ __forceinline__ WaveTimer(unsigned, unsigned) {}
	__device__ __forceinline__ void done(unsigned) const {}
};
#endif
// records of segment `wave` of a queue that a launch takes, given the segment's count
__device__ __forceinline__ unsigned taken(unsigned count, unsigned seg_cap, unsigned first, unsigned take) {
	unsigned c = min(count, seg_cap);
	c = c > first ? c - first : 0;
	return take ? min(c, take) : c;
}
// 64-record steps of wave `wave` in a launch; n = records it takes of its segment (depth > 0)
template <bool PRIMARY>
__device__ __forceinline__ unsigned wave_steps(const BounceArgs& A, unsigned wave, unsigned& n) {
	if (PRIMARY) {
		const unsigned units = (A.batch + 63) >> 6;  // dealt round-robin: wave w takes units w, w + n_seg, ...
		n = 0;
		return wave < units ? (units - wave + A.n_seg - 1) / A.n_seg : 0;
	}
	n = taken(A.in_count[wave], A.seg_cap, A.in_first, A.in_take);
	return (n + 63) >> 6;
}
// lane's record in step k: `ref` = sample index in the batch (depth 0) or slot in `in`; false = no record
template <bool PRIMARY>
__device__ __forceinline__ bool step_record(const BounceArgs& A, unsigned wave, unsigned n, unsigned k, unsigned lane, unsigned& ref) {
	if (PRIMARY) {
		ref = (wave + k * A.n_seg) * 64 + lane;
		return ref < A.batch;
	}
	const unsigned r = k * 64 + lane;
	ref = wave * A.seg_cap + A.in_first + r;
	return r < n;
}

// the input ray behind `ref`: generated (depth 0, framework/renderer.cpp:76-82) or read from `in`
template <bool PRIMARY>
__device__ __forceinline__ void input_ray(const BounceArgs& A, unsigned ref, f3& o, f3& d, f3& w, unsigned& pix) {
	if (PRIMARY) {
		unsigned dp, s;
		udivmod(A.first_s + ref, A.ns, A.inv_ns, dp, s);
		pix = A.first_pix + dp;
		s += A.s0;
		make_primary(A.cam, pix, s, o, d);
		w = mk(1, 1, 1);
	} else {
		const float4 a = A.in.a[ref], b = A.in.b[ref];
		const float2 c = A.in.c[ref];
		o = mk(a.x, a.y, a.z);
		d = mk(a.w, b.x, b.y);
		w = mk(b.z, b.w, c.x);
		pix = __float_as_uint(c.y);
	}
}

// wave64 ballot compaction of up to two records per lane into the wave's own segment of a
// queue: the records of a lane stay adjacent; `tail` (the same in every lane) is the wave's
// count so far.  A record that would not fit is dropped and flagged (the host's sizing rules
// that out; kept as a tripwire).
template <bool WITH_C = true>
__device__ __forceinline__ void emit_pair(const RayQueue& Q, size_t seg_base, unsigned seg_cap, unsigned& tail, bool live1, const float4 r1[3],
                                          bool live2, const float4 r2[3], Counters* counters) {
	const unsigned long long m1 = __ballot(live1), m2 = __ballot(live2);
	const unsigned total = __builtin_popcountll(m1) + __builtin_popcountll(m2);
	if (!total) return;
	const unsigned lane = lane_id();
	const unsigned long long lt = lanes_below(lane);
	unsigned my = tail + __builtin_popcountll(m1 & lt) + __builtin_popcountll(m2 & lt);
	if (live1) {
		if (my < seg_cap) {
			Q.a[seg_base + my] = r1[0];
			Q.b[seg_base + my] = r1[1];
			if (WITH_C) Q.c[seg_base + my] = make_float2(r1[2].x, r1[2].y);
		}
		++my;
	}
	if (live2 && my < seg_cap) {
		Q.a[seg_base + my] = r2[0];
		Q.b[seg_base + my] = r2[1];
		if (WITH_C) Q.c[seg_base + my] = make_float2(r2[2].x, r2[2].y);
	}
	tail += total;
	if (lane == 0 && tail > seg_cap) atomicAdd(&counters->overflow, 1ull);
}

__device__ __forceinline__ void pack_child(const Child& c, unsigned pix, float4 r[3]) {
	r[0] = make_float4(c.o.x, c.o.y, c.o.z, c.d.x);
	r[1] = make_float4(c.d.y, c.d.z, c.w.x, c.w.y);
	r[2] = make_float4(c.w.z, __uint_as_float(pix), 0.f, 0.f);
}

// a wave's share of the work counters, added once at its end (layout: kStatWords)
__device__ __forceinline__ unsigned long long* stat_slot(const BounceArgs& A, unsigned wave) {
	const unsigned slot = A.depth < kMaxDepthSlots ? A.depth : kMaxDepthSlots - 1;
	return A.stat + ((size_t) slot * kStatWaves + (wave % kStatWaves)) * kStatWords;
}
__device__ __forceinline__ void stat_add(unsigned long long* st, int word, unsigned per_lane) {
	unsigned long long v = per_lane;
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	if (lane_id() == 0 && v) atomicAdd(st + word, v);
}
template <class TC>
__device__ __forceinline__ void stat_add_tests(unsigned long long* st, int first_word, const TC& tc) {
	for (int k = 0; k < kTestKinds; ++k) stat_add(st, first_word + k, tc.get(k));
}
template <>
__device__ __forceinline__ void stat_add_tests<NoCount>(unsigned long long*, int, const NoCount&) {}

// ---- fused pipeline: one kernel per depth -----------------------------------------------
template <bool PRIMARY, bool INTPOW>
__global__ __launch_bounds__(kWalkBlock, PTX_MIN_WAVES) void k_bounce(BounceArgs A) {
	extern __shared__ int lds_stack[];  // walk_lds_bytes
	int* stk = lds_stack + threadIdx.x;
	const unsigned lane = lane_id();
	const unsigned wave = wave_id<kWalkBlock>();
	unsigned n;
	const unsigned steps = wave_steps<PRIMARY>(A, wave, n);
	const size_t seg_base = (size_t) wave * A.seg_cap;
	unsigned n_out = 0;
	unsigned cnt_trace = 0, cnt_shadow = 0, cnt_hits = 0;

	for (unsigned k = 0; k < steps; ++k) {
		unsigned ref;
		const bool active = step_record<PRIMARY>(A, wave, n, k, lane, ref);
		f3 o = {0, 0, 0}, d = {0, 0, -1}, w = {1, 1, 1};
		unsigned pix = 0;
		if (active) input_ray<PRIMARY>(A, ref, o, d, w, pix);
		Hit h;
		h.id = -1;
		if (active) {
			trace_closest(A.S, o, d, stk, h);
			++cnt_trace;
		}
		bool hit = active && h.id >= 0;
		f3 contrib = {0, 0, 0};
		Child refl, refr;
		refl.live = refr.live = false;
		if (hit) {
			Surface sf = surface_of(A.S, o, d, h);
			contrib = shade<INTPOW>(A.S, sf, h.id, w, A.last_depth != 0, A.all_children != 0, stk, refl, refr, cnt_shadow);
			++cnt_hits;
		}
		accumulate(A, pix, contrib, hit);
		float4 r1[3], r2[3];
		pack_child(refl, pix, r1);
		pack_child(refr, pix, r2);
		emit_pair(A.out, seg_base, A.seg_cap, n_out, refl.live, r1, refr.live, r2, A.counters);
	}
	if (lane == 0) A.out_count[wave] = n_out;
	unsigned long long* st = stat_slot(A, wave);
	stat_add(st, 0, cnt_trace);
	stat_add(st, 1, cnt_shadow);
	stat_add(st, 2, cnt_hits);
}

// ---- staged pipeline: closest hit / shading / shadow rays as three lean kernels ---------
// k_trace needs no shading state and k_shade no traversal state, so each fits twice the
// waves per SIMD of the fused kernel.  k_trace leaves a compacted list of 16-byte hit
// records (most primary rays of a typical frame hit nothing), so k_shade runs full waves of
// hits only and fetches (or, at depth 0, regenerates) just the rays that hit.
// Only some of a step's 64 rays get as far as a mesh; the others are settled by the leaves of
// frame 0 or miss every mesh's bounds.  Walking a BVH with the few that remain leaves most
// lanes idle (measured: 13 of 64 on the incoherent depths), so unless most of the step goes
// on (kWalkNow lanes) the wave parks those rays - their slot in the queue and the best hit so
// far - in a small ring in LDS and walks once 64 of them have gathered, with every lane busy
// at the start.
// `where` of a hit record: root leaf = its slot in root_prims (>= 0), mesh triangle = -(frame); with a split
// root the octant levels above the child ride in the bits from kExtraShift up (0 otherwise: the plain numbers)
constexpr int kExtraShift = 24;
constexpr int kMaxExtra = 63;
__device__ __forceinline__ int hit_where(const Hit& h) {
	const int v = (h.frame == 0 ? h.slot : h.frame) | (h.extra << kExtraShift);
	return h.frame == 0 ? v : -v;
}
// (measured again with 32 768 segments per launch, ms per image on the mesh scene: walk on the spot from 24 lanes on 280.3,
// from 40 on 278.8, from 56 on 280.1, never 289.9; a ring of 192 rays 285.9)
// (96 rays, not 128: the closest-hit kernel of the deeper depths then fits 24 waves into a compute unit instead of 22,
// 65.7 -> 62.3 ms per image; 80: the same.  A step that would overflow the ring walks what is parked first.)
#ifndef PTX_RING
#define PTX_RING 96
#endif
constexpr unsigned kRing = PTX_RING;    // parked rays per wave
// dynamic LDS of a launch of a kernel that walks: a stack of `entries` per lane, and for the closest-hit kernel of the
// staged pipeline the rings in front of it
constexpr size_t walk_lds_bytes(int entries, bool with_ring) {
	return (size_t) (entries < 2 ? 2 : entries) * kWalkBlock * sizeof(int) + (with_ring ? (kWalkBlock / 64) * kRing * sizeof(float4) : 0);
}
#ifndef PTX_WALK_NOW
#define PTX_WALK_NOW 40
#endif
constexpr unsigned kWalkNow = PTX_WALK_NOW;  // lanes of a step that reach a mesh from which the step walks on the spot
#ifndef PTX_TRACE_WAVES
#define PTX_TRACE_WAVES 4  // (the compiler takes 78 registers unasked: six waves; held to 72 for a seventh it spills: +11 % on that kernel)
#endif
#ifndef PTX_TRACE_WAVES_RB
#define PTX_TRACE_WAVES_RB 4
#endif
#ifndef PTX_PRIMARY_WAVES
#define PTX_PRIMARY_WAVES 5
#endif
template <bool PRIMARY, bool RB, class TC>
__global__ __launch_bounds__(kWalkBlock, (PRIMARY && !RB) ? PTX_PRIMARY_WAVES : ((!PRIMARY && RB) ? PTX_TRACE_WAVES_RB : PTX_TRACE_WAVES)) void k_trace(BounceArgs A_) {
	PTX_ARGS_IN_PLACE(ka);
#define A (*(const BounceArgs*) ka)
	extern __shared__ float4 walk_lds[];  // the rings of the workgroup's waves, then the stack (walk_lds_bytes); primary rays never park: no rings
	int* stk = reinterpret_cast<int*>(walk_lds + (PRIMARY ? 0 : (kWalkBlock / 64) * kRing)) + threadIdx.x;
	volatile float4* ring = walk_lds + (threadIdx.x >> 6) * kRing;
	const unsigned lane = lane_id();
	const unsigned wave = PRIMARY ? wave_id<kWalkBlock>() : segment_of(A.order, wave_id<kWalkBlock>());
	const WaveTimer timer(wave_id<kWalkBlock>(), wave);
	unsigned n;
	const unsigned steps = wave_steps<PRIMARY>(A, wave, n);
	if (!PRIMARY && A.flag_remainder && A.in_take && lane == 0 && min(A.in_count[wave], A.seg_cap) > A.in_first + A.in_take) atomicAdd(&A.counters->remainder, 1ull);
	float4* const my_hits = A.hits + (size_t) wave * A.seg_cap;
	unsigned n_hit = 0;
	unsigned cnt_trace = 0;
	TC tc;
	unsigned parked = 0;  // the same in every lane
	// the hit records of the lanes with `hit`, compacted, appended to the wave's hit segment
	auto emit = [&](bool hit, const Hit& h, unsigned ref) {
		const unsigned long long m = __ballot(hit);
		if (!m) return;
		const unsigned my = n_hit + __builtin_popcountll(m & lanes_below(lane));
		if (hit && my < A.seg_cap) my_hits[my] = make_float4(h.t, __int_as_float(h.id), __int_as_float(hit_where(h)), __uint_as_float(ref));
		n_hit += __builtin_popcountll(m);  // (never beyond seg_cap: a ray leaves at most one hit)
	};
	// the mesh walks of the `count` most recently parked rays
	auto walk_parked = [&](unsigned count) {
		Hit h;
		h.id = -1;
		h.frame = 0;
		h.slot = 0;
		h.extra = 0;
		h.t = kInf;
		unsigned ref = 0;
		if (lane < count) {
			const volatile float4* rq = ring + (parked - count + lane);
			h.t = rq->x;
			h.id = __float_as_int(rq->y);
			h.slot = __float_as_int(rq->z);
			if (RB) {
				h.extra = h.slot >> kExtraShift;
				h.slot &= (1 << kExtraShift) - 1;
			}
			ref = __float_as_uint(rq->w);
			f3 o, d, w;
			unsigned pix;
			input_ray<PRIMARY>(A, ref, o, d, w, pix);
			const RayPre pre = ray_pre(A.S, o, d);
			closest_frames<false, RB>(A.S, o, d, pre, 0u, stk, h, tc);
			if (RB) hit_key_to_id(A.S, h);
		}
		parked -= count;
		emit(h.id >= 0, h, ref);
	};
	// depth > 0: planes a, b of the ray records (origin, direction; the weight and the pixel are
	// the shading kernel's business), fetched one step ahead; the hit records of a step are stored
	// at the top of the next one (see pipeline_wait)
	// (a lane without a record reads its segment's first slot instead: an unconditional load goes
	// straight into the registers the next step reads, a conditional one would be copied there
	// behind a wait)
	float4 next_a = make_float4(0.f, 0.f, 0.f, 0.f), next_b = next_a;
	auto fetch = [&](unsigned k) {
		unsigned ref;
		if (!step_record<PRIMARY>(A, wave, n, k, lane, ref)) ref = wave * A.seg_cap;
		next_a = A.in.a[ref];
		next_b = A.in.b[ref];
	};
	float4 pend_rec = make_float4(0.f, 0.f, 0.f, 0.f);
	unsigned pend_at = 0;
	bool pend = false;
	if (!PRIMARY && steps) fetch(0);
	for (unsigned k = 0; k < steps; ++k) {
		PTX_ARGS_AGAIN(ka);
		unsigned ref;
		const bool active = step_record<PRIMARY>(A, wave, n, k, lane, ref);
		Hit h;
		h.id = -1;
		h.frame = 0;
		h.slot = 0;
		h.extra = 0;
		h.t = kInf;
		f3 o = {0, 0, 0}, d = {0, 0, 1}, w;
		unsigned pix, frames = 0;
		if (!PRIMARY) {
			pipeline_wait();
			const float4 a = next_a, b = next_b;
			if (pend) my_hits[pend_at] = pend_rec;
			pend = false;
			fetch(k + 1 < steps ? k + 1 : k);
			if (active) {
				o = mk(a.x, a.y, a.z);
				d = mk(a.w, b.x, b.y);
			}
		} else if (active) {
			input_ray<PRIMARY>(A, ref, o, d, w, pix);
		}
		RayPre pre = ray_pre(A.S, o, d);
		const bool nan_ray = !PRIMARY && active && nan_direction(d);  // (see nan_ray_hit)
		if (active) {
			if (!nan_ray && closest_root<RB>(A.S, o, d, pre, stk, h, tc)) frames = frame_mask<RB>(A.S, o, d, pre, tc);
			++cnt_trace;
		}
		bool park = frames != 0;
		const unsigned long long m = __ballot(park);
		const unsigned np = __builtin_popcountll(m);
		// most of the wave goes on to a mesh (or these are primary rays, neighbours in the image):
		// walk here, the rays are in registers
		const bool walk_now = PRIMARY || np >= kWalkNow;
		if (walk_now) {
			if (park) closest_frames<true, RB>(A.S, o, d, pre, frames, stk, h, tc);
			park = false;
		}
		if (RB && !park) hit_key_to_id(A.S, h);  // (a parked ray keeps its key until its mesh walks are done)
		if (nan_ray) nan_ray_hit(A.S, h);
		if (PRIMARY) {
			emit(h.id >= 0 && !park, h, ref);
		} else {  // the step's own hits: placed now, stored at the top of the next step
			const bool hit = h.id >= 0 && !park;
			const unsigned long long hm = __ballot(hit);
			pend = hit;
			pend_at = n_hit + __builtin_popcountll(hm & lanes_below(lane));
			pend_rec = make_float4(h.t, __int_as_float(h.id), __int_as_float(hit_where(h)), __uint_as_float(ref));
			n_hit += __builtin_popcountll(hm);
		}
		if (np && !walk_now) {
			if (parked + np > kRing) walk_parked(parked < 64 ? parked : 64);
			if (park) {
				volatile float4* rq = ring + (parked + __builtin_popcountll(m & lanes_below(lane)));
				rq->x = h.t;
				rq->y = __int_as_float(h.id);
				rq->z = __int_as_float(RB ? (h.slot | (h.extra << kExtraShift)) : h.slot);
				rq->w = __uint_as_float(ref);
			}
			parked += np;
			if (parked >= 64) walk_parked(64);
		}
	}
	if (pend) my_hits[pend_at] = pend_rec;
	if (parked) walk_parked(parked);
	if (lane == 0) A.hit_count[wave] = n_hit;
	unsigned long long* st = stat_slot(A, wave);
	stat_add(st, 0, cnt_trace);
	stat_add_tests(st, 3, tc);
	timer.done(PRIMARY ? steps * 64 : n);
#undef A
}

// The shading kernel as a pipeline (see pipeline_wait): at the TOP of step k the wave
//   - writes out what step k-1 produced: its child rays and shadow-ray records, which step k-1
//     left compacted in a staging area in LDS, go to the queues with contiguous 16-byte stores,
//     and its contribution is added to the pixels;
//   - requests what step k+1 will read: the ray each hit of step k+1 belongs to (a gather through
//     the hit record, which arrived a step earlier), the per-triangle record of a mesh hit, and
//     the hit records of step k+2;
// and then shades step k from registers.  The small tables every hit needs (leaf, matrices,
// material) are read from copies in LDS when the scene's fit (LT), so that nothing in the body
// of a step waits for global memory.
constexpr int kLdsRoot = 16;    // frame-0 leaves, mesh frames and materials whose tables fit the LDS copies
constexpr int kLdsFrames = 8;
constexpr int kLdsMats = 32;
constexpr int kStageLights = 2;  // lights whose shadow-ray records go through the staging area (further lights: stored directly)
// Staging area of one wave, cut from the launch's dynamic LDS: room for the records ONE step can
// produce on this scene (BounceArgs::stage_children = 64 or 128 child rays, stage_lights x 64
// shadow-ray records), so that a scene with one light and one-child materials leaves the LDS to
// a fourth workgroup per CU.
struct ShadeStage {
	float4 *child_a, *child_b;
	float2* child_c;
	float4 *shadow_a, *shadow_b;
};
__host__ __device__ constexpr unsigned shade_stage_f4(unsigned children, unsigned lights) { return children * 2 + children / 2 + lights * 64 * 2; }  // float4 per wave
template <bool LT>
struct ShadeLds {
	DevRootPrim prim[LT ? kLdsRoot : 1];
	DevRootShade shade[LT ? kLdsRoot : 1];
	DevXform xf[LT ? kLdsRoot : 1];
	Mat16 fwinv[LT ? kLdsFrames : 1], fw[LT ? kLdsFrames : 1];
	float4 mats[LT ? kLdsMats * 4 : 1];
};
// Fills the LDS copies: the six tables as one list of 16-byte pieces dealt to the threads of the
// workgroup, all loads of a round of pieces in flight together (a loop per table waited for
// memory six times in a row).
template <bool LT>
__device__ __forceinline__ void fill_shade_lds(ShadeLds<LT>& lds, const DevScene& S) {
	const int np[6] = {S.n_root_prims * (int) (sizeof(DevRootPrim) / 16), S.n_root_prims * (int) (sizeof(DevRootShade) / 16),
	                   S.n_root_prims * (int) (sizeof(DevXform) / 16),    S.n_frames * (int) (sizeof(Mat16) / 16),
	                   S.n_frames * (int) (sizeof(Mat16) / 16),           S.n_materials * 4};
	const float4* const src[6] = {reinterpret_cast<const float4*>(S.root_prims), reinterpret_cast<const float4*>(S.root_shade),
	                              reinterpret_cast<const float4*>(S.xforms),     reinterpret_cast<const float4*>(S.frame_winv),
	                              reinterpret_cast<const float4*>(S.frame_w),    S.mat_table};
	float4* const dst[6] = {reinterpret_cast<float4*>(lds.prim),  reinterpret_cast<float4*>(lds.shade), reinterpret_cast<float4*>(lds.xf),
	                        reinterpret_cast<float4*>(lds.fwinv), reinterpret_cast<float4*>(lds.fw),    lds.mats};
	const int total = np[0] + np[1] + np[2] + np[3] + np[4] + np[5];
	for (int p0 = 0; p0 < total; p0 += 2 * kShadeBlock) {
		float4 v[2];
		float4* to[2] = {nullptr, nullptr};
#pragma unroll
		for (int r = 0; r < 2; ++r) {
			int p = p0 + r * kShadeBlock + (int) threadIdx.x;
			v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (p < total) {
				int t = 0;
				while (p >= np[t]) p -= np[t++];
				v[r] = src[t][p];
				to[r] = dst[t] + p;
			}
		}
#pragma unroll
		for (int r = 0; r < 2; ++r)
			if (to[r]) *to[r] = v[r];
	}
}

#ifndef PTX_SHADE0_WAVES
#define PTX_SHADE0_WAVES 4
#endif
template <bool PRIMARY, bool INTPOW, bool LT>
#ifndef PTX_SHADE_GENERAL_WAVES
#define PTX_SHADE_GENERAL_WAVES 3
#endif
__global__ __launch_bounds__(kShadeBlock, (PRIMARY && INTPOW) ? PTX_SHADE0_WAVES : (INTPOW ? 3 : PTX_SHADE_GENERAL_WAVES)) void k_shade(BounceArgs A_) {
	PTX_ARGS_IN_PLACE(ka);
#define A (*(const BounceArgs*) ka)
	__shared__ ShadeLds<LT> lds;
	extern __shared__ float4 stage_mem[];
	if (LT) {
		fill_shade_lds(lds, A.S);
		__syncthreads();
	}
	const ShadeTables T = LT ? ShadeTables{lds.prim, lds.shade, lds.xf, lds.fwinv, lds.fw, lds.mats, A.S.root_winv, A.S.leaf_winv, nullptr} : device_tables(A.S);
	ShadeStage stage;
	{
		float4* p = stage_mem + (threadIdx.x >> 6) * shade_stage_f4(A.stage_children, A.stage_lights);
		stage.child_a = p;
		stage.child_b = p + A.stage_children;
		stage.child_c = reinterpret_cast<float2*>(p + 2 * A.stage_children);
		stage.shadow_a = p + 2 * A.stage_children + A.stage_children / 2;
		stage.shadow_b = stage.shadow_a + 64 * A.stage_lights;
	}
	const unsigned lane = lane_id();
	const unsigned wave = segment_of(A.order, wave_id<kShadeBlock>());
	const WaveTimer timer(wave_id<kShadeBlock>(), wave);
	const unsigned n = min(A.hit_count[wave], A.seg_cap);
	const unsigned steps = (n + 63) >> 6;
	const size_t seg_base = (size_t) wave * A.seg_cap;
	const float4* const my_hits = A.hits + seg_base;
	unsigned n_out = 0, n_shadow = 0;
	unsigned cnt_shadow = 0, cnt_hits = 0;

	// ---- requests (unconditional loads: a lane without a record reads a harmless place, see k_trace)
	auto fetch_hit = [&](unsigned k) {
		const unsigned r = k * 64 + lane;
		float4 hr = my_hits[r < n ? r : 0u];
		if (!(r < n)) hr = make_float4(0.f, __int_as_float(-1), 0.f, __uint_as_float(PRIMARY ? 0u : (unsigned) seg_base));
		return hr;
	};
	struct RayRec {
		float4 a, b;
		float2 c;
	};
	auto fetch_ray = [&](const float4& hr, RayRec& r) {  // the ray behind a hit record (depth > 0)
		if (PRIMARY) return;
		const unsigned ref = __float_as_uint(hr.w);
		r.a = A.in.a[ref];
		r.b = A.in.b[ref];
		r.c = A.in.c[ref];
	};
	auto fetch_info = [&](const float4& hr) {  // prim_info of the hit prim (any valid index for a lane without a hit)
		const int id = __float_as_int(hr.y);
		return A.S.prim_info[id < 0 ? 0 : id];
	};

	// ---- what a step leaves for the top of the next one
	unsigned pend_children = 0, pend_child_at = 0;  // staged child rays and their place in this wave's segment of `out`
	unsigned pend_shadows = 0, pend_shadow_at = 0;
	unsigned pend_pix = 0;
	f3 pend_contrib = {0, 0, 0};
	bool pend_hit = false;
	auto flush = [&]() {
		for (unsigned i = lane; i < pend_children; i += 64) {
			const size_t j = seg_base + pend_child_at + i;
			A.out.a[j] = stage.child_a[i];
			A.out.b[j] = stage.child_b[i];
			A.out.c[j] = stage.child_c[i];
		}
		for (unsigned i = lane; i < pend_shadows; i += 64) {
			const size_t j = seg_base + pend_shadow_at + i;
			A.shadow.a[j] = stage.shadow_a[i];
			A.shadow.b[j] = stage.shadow_b[i];
		}
		accumulate(A, pend_pix, pend_contrib, pend_hit);
		pend_children = pend_shadows = 0;
		pend_hit = false;
	};

	if (steps == 0) {
		if (lane == 0) {
			A.out_count[wave] = 0;
			A.shadow_count[wave] = 0;
		}
		timer.done(0);
		return;
	}
	float4 hit_next = fetch_hit(0);  // hit records of the step after the current one
	pipeline_wait();
	RayRec ray_next;
	ray_next.a = ray_next.b = make_float4(0.f, 0.f, 0.f, 0.f);
	ray_next.c = make_float2(0.f, 0.f);
	fetch_ray(hit_next, ray_next);
	float4 info_next = fetch_info(hit_next);
	float4 hit_cur = hit_next;
	hit_next = fetch_hit(1);
	for (unsigned k = 0; k < steps; ++k) {
		PTX_ARGS_AGAIN(ka);
		pipeline_wait();
		const float4 hr = hit_cur, info = info_next;
		const RayRec ray = ray_next;
		hit_cur = hit_next;
		// ---- the burst: results of step k-1 out, records of steps k+1 / k+2 requested
		flush();
		fetch_ray(hit_cur, ray_next);
		info_next = fetch_info(hit_cur);
		hit_next = fetch_hit(k + 2);
		// ---- step k
		const bool hit = __float_as_int(hr.y) >= 0;
		f3 o = {0, 0, 0}, d = {0, 0, -1}, w = {1, 1, 1};
		unsigned pix = 0;
		Surface sf;
		sf.pos = sf.n = sf.rdir = mk(0, 0, 0);
		f3 kd = {0, 0, 0}, ks = {0, 0, 0}, ws = {0, 0, 0};
		float mm = 0.0f, scale0 = 0.0f, k1 = 1.0f;
		f3 contrib = {0, 0, 0};
		Child refl, refr;
		refl.live = refr.live = false;
		if (hit) {
			Hit h;
			h.t = hr.x;
			h.id = __float_as_int(hr.y);
			int where = __float_as_int(hr.z);
			h.extra = 0;
			if (A.S.n_guards) {  // split root: octant levels in the high bits
				const int v = where < 0 ? -where : where;
				h.extra = v >> kExtraShift;
				where = where < 0 ? -(v & ((1 << kExtraShift) - 1)) : (v & ((1 << kExtraShift) - 1));
			}
			h.frame = where < 0 ? -where : 0;
			h.slot = where < 0 ? 0 : where;
			if (PRIMARY) {
				input_ray<true>(A, __float_as_uint(hr.w), o, d, w, pix);
			} else {
				o = mk(ray.a.x, ray.a.y, ray.a.z);
				d = mk(ray.a.w, ray.b.x, ray.b.y);
				w = mk(ray.b.z, ray.b.w, ray.c.x);
				pix = __float_as_uint(ray.c.y);
			}
			sf = surface_of<PRIMARY, !LT>(T, o, d, h, info);
			for (int k = 0; k < h.extra; ++k) sf.n = up_one_level(sf.n);  // octants of a split root above the child (composite.cpp:100)
			const Material M = material_of(T, h, info);
			bounce_terms(sf, M, w, A.last_depth != 0, A.all_children != 0, refl, refr, k1);
			kd = M.kd;
			ks = M.ks;
			mm = M.m;
			scale0 = M.opacity;
			ws = w;
			// ambient part of phong_color (renderer.cpp:140), scaled like the whole local colour
			contrib = cmul(w, (cmul(ld3(A.S.ambient), M.ka) * M.opacity) * k1);
			++cnt_hits;
		}
		// one shadow ray per light whose term can add something; its record carries the finished
		// contribution weight * ((diffuse + specular) * opacity * k1)
		pend_shadow_at = n_shadow;
		for (int l = 0; l < A.S.n_lights; ++l) {
			bool want = false;
			float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
			if (hit) {
				LightTerm L;
				want = light_term<INTPOW>(A.S, l, sf, kd, ks, mm, L);
				f3 term = L.diffuse;
				if (L.has_spec) term = term + L.spec;
				const f3 c = cmul(ws, (term * scale0) * k1);
				// (k_shadow forms the direction and distance to the light again, with the same operations)
				ra = make_float4(sf.pos.x, sf.pos.y, sf.pos.z, __uint_as_float(pix));
				rb = make_float4(c.x, c.y, c.z, __int_as_float(l));
			}
			if (want) ++cnt_shadow;
			const unsigned long long m = __ballot(want);
			const unsigned cnt = __builtin_popcountll(m);
			if (!cnt) continue;
			const unsigned my = __builtin_popcountll(m & lanes_below(lane));
			if (l < (int) A.stage_lights) {
				if (want) {
					stage.shadow_a[pend_shadows + my] = ra;
					stage.shadow_b[pend_shadows + my] = rb;
				}
				pend_shadows += cnt;
			} else if (want && n_shadow + my < A.seg_cap) {  // (behind the staged ones of this step in the segment: their place is reserved)
				A.shadow.a[seg_base + n_shadow + my] = ra;
				A.shadow.b[seg_base + n_shadow + my] = rb;
			}
			n_shadow += cnt;
		}
		{
			const unsigned long long m1 = __ballot(refl.live), m2 = __ballot(refr.live);
			const unsigned long long lt = lanes_below(lane);
			unsigned my = __builtin_popcountll(m1 & lt) + __builtin_popcountll(m2 & lt);
			float4 r[3];
			if (refl.live) {
				pack_child(refl, pix, r);
				stage.child_a[my] = r[0];
				stage.child_b[my] = r[1];
				stage.child_c[my] = make_float2(r[2].x, r[2].y);
				++my;
			}
			if (refr.live) {
				pack_child(refr, pix, r);
				stage.child_a[my] = r[0];
				stage.child_b[my] = r[1];
				stage.child_c[my] = make_float2(r[2].x, r[2].y);
			}
			pend_children = __builtin_popcountll(m1) + __builtin_popcountll(m2);
			pend_child_at = n_out;
			n_out += pend_children;
		}
		// the host's sizing rules an overflow out; kept as a tripwire: what would not fit is dropped and flagged
		if (n_out > A.seg_cap || n_shadow > A.seg_cap) {
			if (lane == 0) atomicAdd(&A.counters->overflow, 1ull);
			pend_children = pend_child_at < A.seg_cap ? min(pend_children, A.seg_cap - pend_child_at) : 0u;
			pend_shadows = pend_shadow_at < A.seg_cap ? min(pend_shadows, A.seg_cap - pend_shadow_at) : 0u;
		}
		pend_pix = pix;
		pend_contrib = contrib;
		pend_hit = hit;
	}
	pipeline_wait();
	flush();
	if (lane == 0) {
		A.out_count[wave] = n_out;
		A.shadow_count[wave] = n_shadow;
	}
	unsigned long long* st = stat_slot(A, wave);
	stat_add(st, 1, cnt_shadow);
	stat_add(st, 2, cnt_hits);
	timer.done(n);
#undef A
}

// find_light_block for every queued shadow ray (framework/renderer.cpp:148, :168-175); an
// unblocked ray adds its contribution to its pixel
template <bool RB, class TC>
#ifndef PTX_SHADOW_WAVES_RB
#define PTX_SHADOW_WAVES_RB 6  // (the variant for a frame 0 with a tree of its own: 80 registers and 20 bytes of scratch; on a room with 60 shapes in it 167.0 -> 160.3 ms against five waves without scratch)
#endif
#ifndef PTX_SHADOW_WAVES
#define PTX_SHADOW_WAVES 6  // (80 registers instead of the 81 the compiler takes unasked: a sixth wave per SIMD; -6 % under the kernel tracer, nothing without it; 7: the same)
#endif
__global__ __launch_bounds__(kWalkBlock, RB ? PTX_SHADOW_WAVES_RB : PTX_SHADOW_WAVES) void k_shadow(BounceArgs A_) {
	PTX_ARGS_IN_PLACE(ka);
#define A (*(const BounceArgs*) ka)
	extern __shared__ int lds_stack[];  // walk_lds_bytes
	int* stk = lds_stack + threadIdx.x;
	const unsigned lane = lane_id();
	const unsigned wave = segment_of(A.order, wave_id<kWalkBlock>());
	const WaveTimer timer(wave_id<kWalkBlock>(), wave);
	const unsigned n = min(A.shadow_count[wave], A.seg_cap);
	const unsigned steps = (n + 63) >> 6;
	const size_t seg_base = (size_t) wave * A.seg_cap;
	TC tc;
	// records one step ahead, sums one step behind (see pipeline_wait)
	// (a lane without a record reads the segment's first slot: see k_trace)
	float4 next_a = make_float4(0.f, 0.f, 0.f, 0.f), next_b = next_a;
	auto fetch = [&](unsigned k) {
		const unsigned r = k * 64 + lane;
		const size_t j = seg_base + (r < n ? r : 0u);
		next_a = A.shadow.a[j];
		next_b = A.shadow.b[j];
	};
	if (steps) fetch(0);
	unsigned prev_pix = 0;
	f3 prev_c = {0, 0, 0};
	bool prev_lit = false;
	for (unsigned k = 0; k < steps; ++k) {
		PTX_ARGS_AGAIN(ka);
		pipeline_wait();
		const float4 a = next_a, b = next_b;
		accumulate(A, prev_pix, prev_c, prev_lit);
		fetch(k + 1 < steps ? k + 1 : k);
		const bool active = k * 64 + lane < n;
		// the light's position through scalar loads, light by light (a vector load here would make
		// the wave wait for the records just requested)
		const int light = __float_as_int(b.w);
		f3 lpos = {0, 0, 0};
		for (int l = 0; l < A.S.n_lights; ++l) {
			const float4 lp = kload(A.S.lights_k + 2 * l);
			if (light == l) lpos = mk(lp.x, lp.y, lp.z);
		}
		unsigned pix = 0;
		f3 c = {0, 0, 0};
		bool lit = false;
		if (active) {
			pix = __float_as_uint(a.w);
			c = mk(b.x, b.y, b.z);
			// direction and distance to the light exactly as light_term formed them (renderer.cpp:144-147)
			const f3 pos = mk(a.x, a.y, a.z);
			f3 ldir = lpos - pos;
			const float dist = length(ldir);
			ldir = normalize(ldir);
			lit = !trace_any<RB>(A.S, pos, ldir, dist, stk, tc);
		}
		prev_pix = pix;
		prev_c = c;
		prev_lit = lit;
	}
	accumulate(A, prev_pix, prev_c, prev_lit);
	stat_add_tests(stat_slot(A, wave), 3 + kTestKinds, tc);
	timer.done(n);
#undef A
}

// order[] = the runs of four segments (4 q .. 4 q + 3) of one depth sorted by the number of 64-record steps a launch
// takes of them, most first: a counting sort in ONE workgroup, a launch of its own between two others.  Every thread reads
// all its runs' counts at once, every wave counts into bins of its own in LDS and the places of a bin go to the waves in
// turn.  Runs of four because the kernel's time goes with the number of items (the lanes of a wave mostly hold the same
// few step counts and queue for the same words of LDS, on the one compute unit the workgroup runs on: 57 us for 32 768
// single segments) and the order need not be finer: neighbouring segments hold neighbouring pixels' rays.  Runs with
// equal step counts come in any order: the order decides when a segment is worked on, never what is computed.
//   (tried instead: the waves of the shading kernel, which know the counts first, count their segments into the bins
//   with one global atomic each, and a kernel of many workgroups places them: the 32 768 atomics of a launch fall on a
//   handful of cache lines - most segments hold about the same - and cost the shading kernel 4 ms per depth.)
constexpr int kOrderBlock = 1024;
constexpr int kOrderBins = 128;
constexpr int kOrderPer = (int) (kMaxSeg / 4 / kOrderBlock) > 0 ? (int) (kMaxSeg / 4 / kOrderBlock) : 1;  // runs per thread
static_assert(kMaxSeg / 4 <= (unsigned) (kOrderPer * kOrderBlock), "k_order: runs per thread");
__global__ __launch_bounds__(kOrderBlock) void k_order(const unsigned* __restrict__ count, unsigned n_seg, unsigned seg_cap, unsigned first, unsigned take,
                                                       unsigned* __restrict__ order) {
	constexpr int kWaves = kOrderBlock / 64;
	__shared__ unsigned bin[kWaves][kOrderBins];  // per wave: runs per bin, then the first place of the wave's share of the bin
	__shared__ unsigned total[kOrderBins];
	const unsigned t = threadIdx.x, w = t >> 6, lane = t & 63u;
	const unsigned n_runs = n_seg / 4;  // (n_seg is a multiple of four)
	const unsigned max_steps = 4 * ((seg_cap + 63) / 64);
	uint4 raw[kOrderPer];
#pragma unroll
	for (int k = 0; k < kOrderPer; ++k) {
		const unsigned q = (unsigned) k * kOrderBlock + t;
		raw[k] = reinterpret_cast<const uint4*>(count)[q < n_runs ? q : 0u];
	}
	for (unsigned i = t; i < (unsigned) (kWaves * kOrderBins); i += kOrderBlock) (&bin[0][0])[i] = 0;
	__syncthreads();
	unsigned key[kOrderPer];
#pragma unroll
	for (int k = 0; k < kOrderPer; ++k) {
		const unsigned steps = ((taken(raw[k].x, seg_cap, first, take) + 63) >> 6) + ((taken(raw[k].y, seg_cap, first, take) + 63) >> 6) +
		                       ((taken(raw[k].z, seg_cap, first, take) + 63) >> 6) + ((taken(raw[k].w, seg_cap, first, take) + 63) >> 6);
		const unsigned q = max_steps < (unsigned) kOrderBins ? steps : (unsigned) ((unsigned long long) steps * (kOrderBins - 1) / max_steps);
		key[k] = (unsigned) (kOrderBins - 1) - min(q, (unsigned) (kOrderBins - 1));  // most steps first
		if ((unsigned) k * kOrderBlock + t < n_runs) atomicAdd(&bin[w][key[k]], 1u);
	}
	__syncthreads();
	if (t < (unsigned) kOrderBins) {  // a bin's runs over all waves ...
		unsigned sum = 0;
		for (int v = 0; v < kWaves; ++v) sum += bin[v][t];
		total[t] = sum;
	}
	__syncthreads();
	if (t < 64) {  // ... the bins' first places (kOrderBins / 64 bins per lane, then across the lanes) ...
		constexpr int kPer = kOrderBins / 64;
		unsigned v[kPer], mine = 0;
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			v[k] = total[t * kPer + k];
			mine += v[k];
		}
		unsigned incl = mine;
		for (int off = 1; off < 64; off <<= 1) {
			const unsigned o = __shfl_up(incl, off);
			if ((int) lane >= off) incl += o;
		}
		unsigned at = incl - mine;
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			total[t * kPer + k] = at;
			at += v[k];
		}
	}
	__syncthreads();
	if (t < (unsigned) kOrderBins) {  // ... and every wave's share of a bin
		unsigned at = total[t];
		for (int v = 0; v < kWaves; ++v) {
			const unsigned c = bin[v][t];
			bin[v][t] = at;
			at += c;
		}
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < kOrderPer; ++k) {
		const unsigned q = (unsigned) k * kOrderBlock + t;
		if (q < n_runs) order[atomicAdd(&bin[w][key[k]], 1u)] = q;
	}
}

// sums the statistic slots of one depth into the compact block the host reads
__global__ __launch_bounds__(kBlock) void k_stats_reduce(const unsigned long long* __restrict__ stat, Counters* __restrict__ out) {
	__shared__ unsigned long long red[kBlock];
	const unsigned depth = blockIdx.x;
	for (unsigned word = 0; word < 3 + 2 * kTestKinds; ++word) {
		unsigned long long a = 0;
		for (unsigned w = threadIdx.x; w < kStatWaves; w += kBlock) a += stat[((size_t) depth * kStatWaves + w) * kStatWords + word];
		red[threadIdx.x] = a;
		__syncthreads();
		for (unsigned s = kBlock / 2; s > 0; s >>= 1) {
			if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
			__syncthreads();
		}
		if (threadIdx.x == 0) {
			const unsigned long long v = red[0];
			if (word == 0) {
				out->per_depth[depth] = v;
				atomicAdd(&out->trace_rays, v);
			} else if (word == 1) {
				out->per_depth_shadow[depth] = v;
				atomicAdd(&out->shadow_rays, v);
			} else if (word == 2) {
				out->per_depth_hits[depth] = v;
				atomicAdd(&out->shaded_hits, v);
			} else {
				const unsigned k = word - 3;
				out->tests[k / kTestKinds][depth][k % kTestKinds] = v;
			}
		}
		__syncthreads();
	}
}

// scene.root->intersect for arbitrary rays (parity tap + ptx_primary_hits)
struct IntersectArgs {
	DevScene S;
	DevCamera cam;
	const float* o;
	const float* d;   // NULL => primary rays of sample `sample`
	long long n;
	unsigned sample;
	int32_t* prim;
	float* t;
	float* pos;
	float* nrm;
	float* rdir;
};
template <bool RB>
__global__ __launch_bounds__(kWalkBlock) void k_intersect(IntersectArgs A) {
	extern __shared__ int lds_stack[];  // walk_lds_bytes
	int* stk = lds_stack + threadIdx.x;
	for (long long i = (long long) blockIdx.x * kWalkBlock + threadIdx.x; i < A.n; i += (long long) gridDim.x * kWalkBlock) {
		f3 o, d;
		if (A.d) {
			o = ld3(A.o + 3 * i);
			d = ld3(A.d + 3 * i);
		} else {
			make_primary(A.cam, (unsigned) i, A.sample, o, d);
		}
		Hit h;
		trace_closest<RB>(A.S, o, d, stk, h);
		if (A.prim) A.prim[i] = h.id;
		bool hit = h.id >= 0;
		if (A.t) A.t[i] = hit ? h.t : 0.0f;
		if (A.pos || A.nrm || A.rdir) {
			Surface sf;
			sf.pos = sf.n = sf.rdir = mk(0, 0, 0);
			if (hit) sf = surface_of<RB>(A.S, o, d, h);
			if (A.pos) { A.pos[3 * i] = sf.pos.x; A.pos[3 * i + 1] = sf.pos.y; A.pos[3 * i + 2] = sf.pos.z; }
			if (A.nrm) { A.nrm[3 * i] = sf.n.x; A.nrm[3 * i + 1] = sf.n.y; A.nrm[3 * i + 2] = sf.n.z; }
			if (A.rdir) { A.rdir[3 * i] = sf.rdir.x; A.rdir[3 * i + 1] = sf.rdir.y; A.rdir[3 * i + 2] = sf.rdir.z; }
		}
	}
}

// a += b1 + b2 + ... over n floats, in that order: the sum of the per-GPU accumulation buffers, every peer's buffer read where it lies
constexpr int kMaxPeers = 15;
struct PeerList {
	const float* p[kMaxPeers];
	int n;
};
__global__ __launch_bounds__(kBlock) void k_add_peers(float* __restrict__ a, PeerList L, size_t n) {
	const size_t n4 = n / 4;
	float4* a4 = reinterpret_cast<float4*>(a);
	for (size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t) gridDim.x * kBlock) {
		float4 x = a4[i];
		for (int k = 0; k < L.n; ++k) {
			const float4 y = reinterpret_cast<const float4*>(L.p[k])[i];
			x.x += y.x;
			x.y += y.y;
			x.z += y.z;
			x.w += y.w;
		}
		a4[i] = x;
	}
	for (size_t i = n4 * 4 + (size_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t) gridDim.x * kBlock) {
		float x = a[i];
		for (int k = 0; k < L.n; ++k) x += L.p[k][i];
		a[i] = x;
	}
}

// a[i] += (float) sums[i]: a render's own sums into the caller's accumulation buffer, one rounding per pixel and channel
__global__ __launch_bounds__(kBlock) void k_fold(float* __restrict__ a, const double* __restrict__ sums, size_t n) {
	for (size_t i = (size_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t) gridDim.x * kBlock) a[i] += (float) sums[i];
}

// framework/renderer.cpp:86-89, :286-291 and ppmwriter.cpp:36-43
__global__ __launch_bounds__(kBlock) void k_resolve(const float* __restrict__ accum, unsigned w, unsigned h, float inv_spp,
                                                    float* __restrict__ color, uint8_t* __restrict__ rgb8) {
	unsigned n = w * h;
	for (unsigned p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
		unsigned x = p % w, y = p / w;
		float c[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			float v = accum[(size_t) p * 3 + k] * inv_spp;
			v /= v + 1.0f;
			c[k] = v;
		}
		if (color) {
			color[(size_t) p * 3] = c[0];
			color[(size_t) p * 3 + 1] = c[1];
			color[(size_t) p * 3 + 2] = c[2];
		}
		if (rgb8) {
			size_t q = 3 * ((size_t) w * (h - 1 - y) + x);
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				double v = 255.0 * (double) c[k];
				v = (255.0 < v) ? 255.0 : v;  // std::min(255.0 * c, 255.0)
				v = (0.0 < v) ? v : 0.0;      // std::max(0.0, .)
				rgb8[q + k] = (uint8_t) (unsigned int) v;
			}
		}
	}
}

}  // namespace
