// MI355X (gfx950) render path: device-resident scene, wavefront kernels and the C ABI of
// include/ptx.h.  Replaces Renderer::render and everything it reaches
// (framework/renderer.cpp:25-291, composite.cpp:75-103, sphere.cpp:35-52, box.cpp:89-165,
// triangle.cpp:56-88 of the reference).
//
// Structure of one render
//   for every chunk of primary samples (pixel-major, sample-minor, so the 64 lanes of a
//   wave are 64 samples of ONE pixel), for every depth d of the ray tree:
//     k_trace    closest hit of every ray of depth d (d = 0: rays generated in registers, no
//                primary-ray queue; d > 0: read from the queue depth d-1 wrote): frame-0 leaves in
//                lock step (wave-uniform scalar loads, a bounding-box filter in front of each exact
//                test; a BVH over them from 17 leaves on) + one BVH2 walk per mesh frame with a
//                per-lane stack in LDS; the hits are compacted into a list of 16-byte records
//     k_shade    one lane per hit: surface point, Phong terms, <= 2 weighted child rays
//                (reflection / refraction) appended to the next depth's queue, one shadow-ray record
//                per light appended to the shadow queue, ambient term accumulated
//     k_shadow   any-hit walk per shadow ray; an unblocked ray adds its contribution
//     (or k_bounce: all of that in one kernel per depth, the fused pipeline)
//   Queues are cut into wave-private segments: wave w of every launch reads segment w of its input
//   and appends to segment w of its outputs by counting in a register (one wave64 ballot per append,
//   no atomics on queue tails); a wave stores its counts once, at its end, and the next launch's
//   wave w reads them on the device.
//   accumulate   wave reduction per pixel run, then one atomic instruction for (r, g, b), into the
//                render's own sums (doubles); k_fold adds them to the caller's float buffer at the end
//   k_resolve    accum / spp -> tone map -> float colour buffer + flipped 8-bit rows
// The host launches every depth of a chunk back to back; it looks at the device once per chunk (one
// word: did a launch leave rays behind?) only for scenes with two-child materials, else never.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is looked up at run time (rccl())
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "ptx.h"
#include "ptx_bvh.hpp"
#include "ptx_device.hpp"
#include "ptx_kernels.hpp"

namespace {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local std::string g_error;

PtxStatus fail(PtxStatus st, std::string msg) {
	g_error = std::move(msg);
	return st;
}
#define HIP_TRY(expr)                                                                                          \
	do {                                                                                                       \
		hipError_t e_ = (expr);                                                                                \
		if (e_ != hipSuccess) return fail(PTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

bool finite16(const float* m) {
	for (int i = 0; i < 16; ++i)
		if (!std::isfinite(m[i])) return false;
	return true;
}
// rotation/scale part exactly the identity (zeros of either sign), bottom row (0,0,0,1):
// such a matrix maps x to x + t with one rounding per component, like the reference's
// (m0*x + m1*y) + (m2*z + m3*1) does when m0..m2 are unit vectors
bool pure_translation(const float* m) {
	for (int c = 0; c < 3; ++c)
		for (int r = 0; r < 3; ++r)
			if (m[c * 4 + r] != (c == r ? 1.0f : 0.0f)) return false;
	return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f;
}
bool is_identity(const float* m) {
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r)
			if (m[c * 4 + r] != (c == r ? 1.0f : 0.0f)) return false;
	return true;
}

}  // namespace

// Device memory a render works in (everything besides the scene tables, the inputs and the
// accumulation buffer): the scene's own block, or one the caller of the driver supplies.
struct WorkMem {
	float4* qmem = nullptr;        // cut per render into ray queues, a shadow queue and a hit list
	uint64_t qslots = 0;           // its size in 48-byte slots
	unsigned* counts = nullptr;    // records per segment: 3 x (kMaxBounces + 2) rows (rays, shadow rays, hits per depth) of max_seg words
	unsigned max_seg = 0;          // most segments (= waves of a grid) a render may use with this memory
	Counters* counters = nullptr;  // kCounterBytes
	unsigned long long* stat = nullptr;  // kStatBytes
};

// Work memory of ptx_render_accumulate_static, for callers that must not allocate: one array
// in the code object, of which every GPU that loads the library holds its own copy until the
// process ends.  In host code the NAME of the array is only a handle (handing it to a kernel
// compiles and then faults on the device): the array's address on a device comes from
// hipGetSymbolAddress, asked once per device with that device current (scratch_on) and kept;
// everything below uses that pointer.  Layout: counters, work counters, segment counts, queues.
constexpr unsigned kStaticSeg = 2048;  // most segments (waves per grid) of a render in this memory
constexpr size_t kStaticBytes = 128ull << 20;
constexpr size_t kStaticStatOff = kCounterBytes;
constexpr size_t kStaticCountsOff = kStaticStatOff + kStatBytes;
constexpr size_t kStaticQueueOff = (kStaticCountsOff + count_bytes(kStaticSeg) + 255) & ~(size_t) 255;
static_assert(kStaticQueueOff + (32ull << 20) <= kStaticBytes, "static work memory: no room left for the queues");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[kMaxDevices] = {};  // the array's address on each device (0 = not asked yet)

// address of `scratch` on `device`, which must be the current device
static hipError_t scratch_on(int device, unsigned char** out) {
	std::lock_guard<std::mutex> lock(g_scratch_mutex);
	if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
	if (!g_scratch_ptr[device]) {
		void* p = nullptr;
		hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
		if (e != hipSuccess) return e;
		g_scratch_ptr[device] = static_cast<unsigned char*>(p);
	}
	*out = g_scratch_ptr[device];
	return hipSuccess;
}

// The queue block of the scene destroyed last on a device, kept for the next scene created there.  Mapping a block
// of tens of GiB is quick as long as nothing has been written to it; releasing one a render has filled, and
// mapping the next right after, took 0.6 to 1.6 s per call in a program that creates and destroys a scene per image
// (32 GiB).  One block per device is held until ptx_trim() or the end of the process.
struct KeptBlock {
	void* p = nullptr;
	uint64_t bytes = 0;
};
static std::mutex g_kept_mutex;
static KeptBlock g_kept[kMaxDevices];

// a block of at least `bytes` on `device` (the current device): the kept one if it is large enough and not more than
// twice what is asked for, else a new one
static hipError_t take_block(int device, uint64_t bytes, void** out, uint64_t* got) {
	{
		std::lock_guard<std::mutex> lock(g_kept_mutex);
		if (device >= 0 && device < kMaxDevices) {
			KeptBlock& k = g_kept[device];
			if (k.p && k.bytes >= bytes && k.bytes <= 2 * bytes) {
				*out = k.p;
				*got = k.bytes;
				k = KeptBlock{};
				return hipSuccess;
			}
		}
	}
	*got = bytes;
	return hipMalloc(out, bytes);
}
// gives a block back: it becomes the device's kept block unless that one is larger
static void give_block(int device, void* p, uint64_t bytes) {
	void* drop = p;
	{
		std::lock_guard<std::mutex> lock(g_kept_mutex);
		if (device >= 0 && device < kMaxDevices && g_kept[device].bytes < bytes) {
			drop = g_kept[device].p;
			g_kept[device].p = p;
			g_kept[device].bytes = bytes;
		}
	}
	if (drop) (void) hipFree(drop);
}

struct PtxScene {
	int device = 0;
	DevScene dev{};
	PtxCamera camera{};
	std::vector<void*> allocs;
	void* block = nullptr;   // the queue block (work.qmem), handed on to the device's next scene at destruction (give_block)
	uint64_t block_bytes = 0;
	WorkMem work;
	std::vector<unsigned> tails;   // host copy of the tail counters a render reads back (sized once, at creation)
	double* sums = nullptr;        // a render's own sums (see accumulate), folded into the caller's buffer at its end
	size_t sums_len = 0;
	void* undo = nullptr;          // one chunk's pixels of `sums` and the work counters as they were before the chunk (render_accumulate)
	size_t undo_bytes = 0;
	unsigned long long* host_word = nullptr;  // pinned: where the one word a chunk is checked by lands (a copy into pageable memory is staged and slow)
	// buffers of ptx_render_image_multi, kept from call to call: this worker's accumulation buffer; its copy on
	// another GPU that cannot read it in place (multi_stage_device); colour and 8-bit rows of the first worker
	float* multi_accum = nullptr;
	size_t multi_accum_bytes = 0;
	float* multi_stage = nullptr;
	size_t multi_stage_bytes = 0;
	int multi_stage_device = -1;
	float* multi_color = nullptr;
	size_t multi_color_bytes = 0;
	uint8_t* multi_rgb = nullptr;
	size_t multi_rgb_bytes = 0;
	Counters host_counters{};
	bool branching = false;  // some material spawns two children per hit
	bool bouncing = false;   // some material spawns any child
	bool int_pow = true;     // every specular exponent is a whole number in [1, kIntPowMax] (or 0 = no specular term)
	int profiling = 0;       // 0 off, 1 events around every launch, 2 events + the counting twins of the traversal kernels
	std::vector<hipEvent_t> events;      // reused from call to call
	std::vector<unsigned> event_tag;     // launch k: kernel kind and depth ...
	std::vector<unsigned> event_begin;   // ... and the event that marks its begin (its end: the event after that)
	int n_prims = 0, n_frames = 0, n_nodes = 0, max_depth = 0, grid = 0;
	uint64_t device_bytes = 0;

	template <class T>
	PtxStatus upload(std::vector<T> const& v, const T** out) {
		void* p = nullptr;
		size_t bytes = std::max<size_t>(v.size() * sizeof(T), 256);
		HIP_TRY(hipMalloc(&p, bytes));
		allocs.push_back(p);
		device_bytes += bytes;
		HIP_TRY(hipMemset(p, 0, bytes));
		if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
		*out = static_cast<const T*>(p);
		return PTX_OK;
	}
};

namespace {

// The root hands every ray on through its world_transform_inv_, the computed inverse of the identity
// (composite.cpp:83, shape.cpp:3-7), which leaves a position as it is except that a negative zero may turn
// positive.  For primary rays the origin is the camera position: mapped here, once.
void root_sees_camera(const DevScene& S, DevCamera& cam) {
	const f3 p = xform_point(S.root_winv, mk(cam.m[12], cam.m[13], cam.m[14]));
	cam.m[12] = p.x;
	cam.m[13] = p.y;
	cam.m[14] = p.z;
}

void camera_basis(const PtxCamera& c, const PtxRenderParams& rp, DevCamera& out) {
	// framework/renderer.cpp:27-37.  tan is the double overload there; the quotient is
	// formed in double and rounded once.
	float fov_rad = c.fov_x / 180 * 3.14159265f;
	float dist = (rp.width / 2.0f) / ::tan((double) (fov_rad / 2));
	const float* dir = c.direction;
	const float* up = c.up;
	float u[3] = {dir[1] * up[2] - up[1] * dir[2], dir[2] * up[0] - up[2] * dir[0], dir[0] * up[1] - up[0] * dir[1]};
	float v[3] = {u[1] * dir[2] - dir[1] * u[2], u[2] * dir[0] - dir[2] * u[0], u[0] * dir[1] - dir[0] * u[1]};
	float m[16] = {u[0], u[1], u[2], 0, v[0], v[1], v[2], 0, -dir[0], -dir[1], -dir[2], 0, c.position[0], c.position[1], c.position[2], 1};
	std::memcpy(out.m, m, sizeof m);
	out.dist = dist;
	out.aa_unit = 1.0f / rp.aa_steps;
	out.half_w = rp.width * 0.5f;
	out.half_h = rp.height * 0.5f;
	out.width = rp.width;
	out.height = rp.height;
	out.aa_steps = rp.aa_steps;
	out.inv_width = 1.0 / rp.width;
	out.inv_aa = 1.0 / rp.aa_steps;
	out.jitter_seed = rp.jitter_seed;
	out.pad = 0;
}

// Bounding box of a frame-0 leaf in frame-0 space for may_hit_box: the leaf's local box (a
// sphere's is centre +- radius) carried through its matrix corner by corner, then padded by
// 0.01 % + 0.001 so that the filter's own rounding can never reject a ray the exact test
// accepts.  Anything non-finite gives an infinite box (never rejects).
void bounding_box(const PtxPrim& P, float4& lo, float4& hi) {
	double l[3], h[3];
	for (int k = 0; k < 3; ++k) {
		if (P.kind == PTX_SPHERE) {
			l[k] = (double) P.p[k] - std::fabs((double) P.p[3]);
			h[k] = (double) P.p[k] + std::fabs((double) P.p[3]);
		} else if (P.kind == PTX_BOX) {
			l[k] = std::min<double>(P.p[k], P.p[3 + k]);
			h[k] = std::max<double>(P.p[k], P.p[3 + k]);
		} else {
			l[k] = std::min<double>(P.p[k], std::min<double>(P.p[3 + k], P.p[6 + k]));
			h[k] = std::max<double>(P.p[k], std::max<double>(P.p[3 + k], P.p[6 + k]));
		}
	}
	const float* W = P.world;  // column-major
	const double inf = std::numeric_limits<double>::infinity();
	double wl[3] = {inf, inf, inf}, wh[3] = {-inf, -inf, -inf};
	bool finite = true;
	for (int c = 0; c < 8; ++c) {
		const double x = c & 1 ? h[0] : l[0], y = c & 2 ? h[1] : l[1], z = c & 4 ? h[2] : l[2];
		for (int k = 0; k < 3; ++k) {
			const double v = W[k] * x + W[4 + k] * y + W[8 + k] * z + W[12 + k];
			finite = finite && std::isfinite(v);
			wl[k] = std::min(wl[k], v);
			wh[k] = std::max(wh[k], v);
		}
	}
	float fl[3], fh[3];
	for (int k = 0; k < 3; ++k) {
		const double pad = 1e-4 * std::max(std::fabs(wl[k]), std::fabs(wh[k])) + 1e-3;
		fl[k] = finite ? std::nextafter((float) (wl[k] - pad), -std::numeric_limits<float>::infinity()) : -std::numeric_limits<float>::infinity();
		fh[k] = finite ? std::nextafter((float) (wh[k] + pad), std::numeric_limits<float>::infinity()) : std::numeric_limits<float>::infinity();
	}
	lo = make_float4(fl[0], fl[1], fl[2], 0.f);
	hi = make_float4(fh[0], fh[1], fh[2], 0.f);
}

#ifndef PTX_ROOT_BVH_LEAF
#define PTX_ROOT_BVH_LEAF 1
#endif
constexpr int kRootBvhMin = 16;  // leaves of frame 0 from which they get a BVH of their own

PtxStatus check_params(const PtxRenderParams* p) {
	if (!p) return fail(PTX_ERR_INVALID, "params is NULL");
	if (p->width == 0 || p->height == 0 || p->aa_steps == 0) return fail(PTX_ERR_INVALID, "width, height and aa_steps must be positive");
	if ((uint64_t) p->width * p->height > 0x7fffffffull / 3) return fail(PTX_ERR_INVALID, "image too large");
	if (p->aa_steps > 32768u) return fail(PTX_ERR_INVALID, "aa_steps too large (at most 32768)");
	if (p->sample_begin > p->sample_end || p->sample_end > p->aa_steps * p->aa_steps)
		return fail(PTX_ERR_INVALID, "sample range outside [0, aa_steps^2]");
	if (p->max_ray_bounces > kMaxBounces) return fail(PTX_ERR_UNSUPPORTED, "max_ray_bounces must not exceed 63");
	if ((p->pipeline & ~(uint32_t) (PTX_TRACE_ZERO_WEIGHT | PTX_REDUCE_RCCL)) > PTX_PIPELINE_STAGED) return fail(PTX_ERR_INVALID, "unknown pipeline");
	return PTX_OK;
}

}  // namespace

extern "C" {

const char* ptx_last_error(void) { return g_error.c_str(); }
const char* ptx_version(void) { return "ptx 0.1 (gfx950)"; }

int ptx_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// entries of the traversal stack a launch asks LDS for: what the scene's deepest tree needs
// (PTX_WALK_ENTRIES="a,b,c,d" in the environment: at least that many for the closest-hit kernel of depth 0, of the deeper
// depths, the shadow kernel of depth 0, of the deeper depths - for measurements of the waves a compute unit holds)
static int stack_entries(const PtxScene* sc, int which = -1) {
	static int at_least[4] = {-1, 0, 0, 0};
	if (at_least[0] < 0) {
		at_least[0] = 0;
		if (const char* e = getenv("PTX_WALK_ENTRIES")) sscanf(e, "%d,%d,%d,%d", &at_least[0], &at_least[1], &at_least[2], &at_least[3]);
	}
	return std::max(sc->max_depth, which < 0 ? 0 : at_least[which]);
}

PtxStatus ptx_scene_create(const PtxSceneDesc* D, int device, uint64_t queue_bytes, PtxScene** out) {
	if (!D || !out) return fail(PTX_ERR_INVALID, "desc/out is NULL");
	*out = nullptr;
	if (D->n_frames < 1 || !D->frames) return fail(PTX_ERR_INVALID, "scene needs frame 0 (the root Composite)");
	if (D->n_prims < 0 || D->n_materials < 0 || D->n_lights < 0) return fail(PTX_ERR_INVALID, "negative count");
	if ((D->n_prims && !D->prims) || (D->n_materials && !D->materials) || (D->n_lights && !D->lights))
		return fail(PTX_ERR_INVALID, "NULL array with non-zero count");
	if (D->n_materials > 0xffff) return fail(PTX_ERR_UNSUPPORTED, "more than 65535 materials");
	if (!is_identity(D->frames[0].world) || D->frames[0].parent != -1)
		return fail(PTX_ERR_UNSUPPORTED, "frame 0 must be the untransformed root Composite");
	// frames inside frames: every frame names an earlier one as the frame it lies in, at most kMaxNest deep
	bool nested = false;
	for (int f = 1; f < D->n_frames; ++f) {
		const int p = D->frames[f].parent;
		if (p < 0 || p >= f) return fail(PTX_ERR_INVALID, "frame " + std::to_string(f) + ": the frame it lies in must come earlier in the list (0 = the root)");
		int depth = 1;
		for (int a = p; a > 0; a = D->frames[a].parent) ++depth;
		if (depth > kMaxNest) return fail(PTX_ERR_UNSUPPORTED, "frame " + std::to_string(f) + ": transformed Composites nested deeper than " + std::to_string(kMaxNest));
		nested = nested || p > 0;
	}
	// ---- octants of a split root
	const bool guarded = D->n_guards > 0;
	if (D->n_guards < 0 || D->n_occurrences < 0 || (guarded && (!D->guards || !D->occurrences)))
		return fail(PTX_ERR_INVALID, "guards / occurrences: negative count, or NULL array with guards present");
	if (guarded) {
		if (D->n_frames >= (1 << kExtraShift)) return fail(PTX_ERR_UNSUPPORTED, "a split root with 2^24 frames or more");
		for (int g = 0; g < D->n_guards; ++g) {
			const PtxGuard& G = D->guards[g];
			if (G.parent < -1 || G.parent >= g) return fail(PTX_ERR_INVALID, "guard " + std::to_string(g) + ": the octant above must come earlier in the list (or be -1)");
			const int want = G.parent < 0 ? 2 : D->guards[G.parent].levels + 1;
			if (G.levels != want) return fail(PTX_ERR_INVALID, "guard " + std::to_string(g) + ": levels must be one more than that of the octant above (2 under the root)");
			if (G.levels - 1 > kMaxExtra) return fail(PTX_ERR_UNSUPPORTED, "guard " + std::to_string(g) + ": octants nested deeper than " + std::to_string(kMaxExtra) + " levels");
		}
		for (int k = 0; k < D->n_occurrences; ++k) {
			const PtxOccurrence& O = D->occurrences[k];
			const bool leaf = O.prim >= 0 && O.prim < D->n_prims && O.frame == 0 && D->prims[O.prim].frame == 0;
			const bool mesh = O.prim == -1 && O.frame >= 1 && O.frame < D->n_frames && D->frames[O.frame].parent == 0;
			if (O.guard < -1 || O.guard >= D->n_guards || !(leaf || mesh))
				return fail(PTX_ERR_INVALID, "occurrence " + std::to_string(k) + ": names no octant, or neither a frame-0 leaf nor a frame that lies directly in the root");
		}
	}
	// ---- every index and count the set-up below and the kernels rely on, before anything touches a device
	for (int f = 1; f < D->n_frames; ++f)
		if (D->frames[f].levels_up < (D->frames[f].parent == 0 ? 1 : 0) || D->frames[f].levels_up > 255)
			return fail(PTX_ERR_INVALID, "frame " + std::to_string(f) + ": levels_up out of range");
	for (int i = 0; i < D->n_prims; ++i) {
		const PtxPrim& P = D->prims[i];
		if (P.kind < 0 || P.kind > 2) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": unknown kind");
		if (P.frame < 0 || P.frame >= D->n_frames) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": frame out of range");
		if (P.material < 0 || P.material >= D->n_materials) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": material out of range");
		if (P.levels_in < 0 || P.levels_in > 255) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": levels_in out of range");
	}
	int ndev = ptx_device_count();
	if (ndev <= 0) return fail(PTX_ERR_DEVICE, "no HIP device: this library has no CPU path");
	if (device < 0 || device >= ndev) return fail(PTX_ERR_INVALID, "device index out of range");
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
		return fail(PTX_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only");

	{  // this device's copy of the static work memory (ptx_render_accumulate_static): its address is asked here, once per device
		unsigned char* unused = nullptr;
		HIP_TRY(scratch_on(device, &unused));
	}
	std::unique_ptr<PtxScene> sc(new PtxScene);
	sc->device = device;
	sc->camera = D->camera;
	sc->n_prims = D->n_prims;
	sc->n_frames = D->n_frames;
	sc->grid = prop.multiProcessorCount * 8;

	// ---- materials, lights
	std::vector<float4> mats((size_t) D->n_materials * 4);
	for (int i = 0; i < D->n_materials; ++i) {
		const PtxMaterial& m = D->materials[i];
		mats[4 * i + 0] = make_float4(m.ka[0], m.ka[1], m.ka[2], m.kd[0]);
		mats[4 * i + 1] = make_float4(m.kd[1], m.kd[2], m.ks[0], m.ks[1]);
		mats[4 * i + 2] = make_float4(m.ks[2], m.m, m.glossy, m.opacity);
		mats[4 * i + 3] = make_float4(m.ior, 0, 0, 0);
		bool refl = m.glossy > 0, refr = m.opacity < 1;
		if (refl && refr) sc->branching = true;
		if (refl || refr) sc->bouncing = true;
		if (m.m != 0.0f && !(m.m >= 1.0f && m.m <= kIntPowMax && m.m == std::floor(m.m))) sc->int_pow = false;
	}
	std::vector<float4> lights((size_t) D->n_lights * 2);
	for (int i = 0; i < D->n_lights; ++i) {
		lights[2 * i] = make_float4(D->lights[i].position[0], D->lights[i].position[1], D->lights[i].position[2], 0);
		lights[2 * i + 1] = make_float4(D->lights[i].intensity[0], D->lights[i].intensity[1], D->lights[i].intensity[2], 0);
	}

	// ---- leaves: frame 0 leaves stay a list; every other frame's triangles go into a BVH.
	// A leaf or frame whose transform holds a non-finite number can never be hit by the
	// reference's sphere test and is dropped here for every kind.
	// split root: the occurrences of every child of the root, grouped per child in visiting order, each with
	// its tie-break key = its rank in that order (a mesh occurrence takes one key per prim id of the mesh)
	std::vector<std::vector<int2>> prim_occ(guarded ? D->n_prims : 0), mesh_occ(guarded ? D->n_frames : 0);
	std::vector<int> frame_id_lo(D->n_frames, INT32_MAX), frame_id_hi(D->n_frames, -1);
	for (int i = 0; i < D->n_prims; ++i) {
		const int f = D->prims[i].frame;
		if (f < 0 || f >= D->n_frames) continue;  // (reported below)
		frame_id_lo[f] = std::min(frame_id_lo[f], i);
		frame_id_hi[f] = std::max(frame_id_hi[f], i);
	}
	// (a frame that holds frames: the prim ids of everything below it, one run, as the depth-first listing gives them)
	for (int f = D->n_frames - 1; f >= 1; --f) {
		const int p = D->frames[f].parent;
		if (p <= 0 || frame_id_hi[f] < 0) continue;
		frame_id_lo[p] = std::min(frame_id_lo[p], frame_id_lo[f]);
		frame_id_hi[p] = std::max(frame_id_hi[p], frame_id_hi[f]);
	}
	if (guarded) {
		int64_t key = 0;
		for (int k = 0; k < D->n_occurrences; ++k) {
			const PtxOccurrence& O = D->occurrences[k];
			if (O.prim >= 0) {
				prim_occ[O.prim].push_back(make_int2(O.guard, (int) key));
				key += 1;
			} else {
				mesh_occ[O.frame].push_back(make_int2(O.guard, (int) key));
				key += frame_id_hi[O.frame] >= 0 ? frame_id_hi[O.frame] - frame_id_lo[O.frame] + 1 : 0;
			}
			if (key > (int64_t) 1 << 30) return fail(PTX_ERR_UNSUPPORTED, "a split root whose occurrences span more than 2^30 tie-break keys");
		}
	}
	std::vector<int2> occ, root_occ;
	std::vector<int4> frame_occ(D->n_frames, make_int4(0, 0, 0, 0));
	std::vector<DevRootPrim> root_prims;
	std::vector<DevRootShade> root_shade;
	std::vector<float4> root_bound;
	std::vector<DevXform> xforms;
	std::vector<float4> prim_info((size_t) std::max(D->n_prims, 1));
	std::vector<std::vector<int>> frame_tris(D->n_frames);
	std::vector<std::vector<DevRootPrim>> frame_leaf_prims(D->n_frames);  // leaves of frames other than the root that are not plain triangles
	std::vector<std::vector<DevRootShade>> frame_leaf_shade(D->n_frames);
	std::vector<std::vector<DevXform>> frame_leaf_xforms(D->n_frames);
	std::vector<char> frame_ok(D->n_frames, 1);
	for (int f = 1; f < D->n_frames; ++f) {
		const PtxFrame& F = D->frames[f];
		bool ok = finite16(F.world) && finite16(F.world_inv);
		for (int k = 0; k < 3; ++k) ok = ok && !std::isnan(F.bounds_min[k]) && !std::isnan(F.bounds_max[k]);
		frame_ok[f] = ok && frame_ok[F.parent];  // (a frame inside a dropped frame goes with it)
	}
	for (int i = 0; i < D->n_prims; ++i) {
		const PtxPrim& P = D->prims[i];
		int packed = P.material | (P.levels_in << 16);
		float pw;
		std::memcpy(&pw, &packed, 4);
		{
			// The normal a hit on a triangle reports does not depend on the ray, so the chain the
			// reference applies per hit is applied once here, with the device's own functions:
			// mesh triangle - re-normalised by every Composite between it and the mesh
			// (octants, sub-objects; identity transforms), mapped by the mesh's transform and
			// normalised (composite.cpp:99-100), re-normalised by every Composite above the mesh;
			// root triangle - mapped by its own transform (triangle.cpp:86), then re-normalised
			// by the Composites above it.
			f3 n = mk(P.normal[0], P.normal[1], P.normal[2]);
			if (P.frame > 0) {
				if (P.kind == PTX_TRIANGLE && !is_identity(P.world)) n = xform_dir(P.world, n);  // (a moved triangle inside a frame: its own transform first, triangle.cpp:86)
				for (int k = 0; k < P.levels_in; ++k) n = up_one_level(n);
				// (a frame inside a frame: the same again for every frame up the chain)
				for (int a = P.frame; a > 0; a = D->frames[a].parent) {
					const PtxFrame& F = D->frames[a];
					n = normalize(xform_dir(F.world, n));
					// (split root: once for the root here, once more per octant level above the occurrence
					// of the mesh a ray came through when the hit is shaded)
					for (int k = 0; k < (guarded && F.parent == 0 ? 1 : F.levels_up); ++k) n = up_one_level(n);
				}
			} else if (P.kind == PTX_TRIANGLE) {
				n = xform_dir(P.world, n);
				for (int k = 0; k < (guarded ? 1 : P.levels_in); ++k) n = up_one_level(n);
			}
			prim_info[i] = make_float4(n.x, n.y, n.z, pw);
		}
		bool finite = finite16(P.world) && finite16(P.world_inv);
		for (int k = 0; k < 9; ++k) finite = finite && std::isfinite(P.p[k]);
		if (P.frame == 0) {
			if (P.levels_in < 1) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": a root leaf sits under at least the root (levels_in >= 1)");
			if (!finite) continue;
			DevRootPrim R{};
			DevRootShade Q{};
			const bool plain = pure_translation(P.world) && pure_translation(P.world_inv);
			R.kind_flags = P.kind | (plain ? kPlain : 0);
			for (int k = 0; k < 3; ++k) {
				R.tinv[k] = P.world_inv[12 + k];
				Q.tw[k] = P.world[12 + k];
			}
			std::memcpy(R.p, P.p, sizeof R.p);
			if (P.kind == PTX_TRIANGLE)
				for (int k = 0; k < 3; ++k) {
					R.p[3 + k] = P.p[3 + k] - P.p[k];  // e1 = v1 - v0, framework/triangle.cpp:58
					R.p[6 + k] = P.p[6 + k] - P.p[k];  // e2 = v2 - v0, :59
				}
			R.id = i;
			{
				DevXform X{};
				std::memcpy(X.winv, P.world_inv, sizeof X.winv);
				std::memcpy(X.w, P.world, sizeof X.w);
				R.xform = (int32_t) xforms.size();
				xforms.push_back(X);
			}
			if (plain && P.kind == PTX_SPHERE) sc->dev.has_plain_sphere = 1;
			Q.n[0] = prim_info[i].x;
			Q.n[1] = prim_info[i].y;
			Q.n[2] = prim_info[i].z;
			Q.levels_in = guarded ? 1 : P.levels_in;
			Q.material = P.material;
			if (guarded) {
				if (prim_occ[i].empty()) return fail(PTX_ERR_INVALID, "prim " + std::to_string(i) + ": a leaf of a split root without an occurrence");
				root_occ.push_back(make_int2((int) occ.size(), (int) prim_occ[i].size()));
				occ.insert(occ.end(), prim_occ[i].begin(), prim_occ[i].end());
			}
			root_prims.push_back(R);
			root_shade.push_back(Q);
			{
				float4 blo, bhi;
				bounding_box(P, blo, bhi);
				root_bound.push_back(blo);
				root_bound.push_back(bhi);
			}
		} else if (P.kind == PTX_TRIANGLE && is_identity(P.world)) {
			if (finite && frame_ok[P.frame]) frame_tris[P.frame].push_back(i);
		} else {
			// a sphere, a box or a moved triangle inside a transformed Composite: kept as a leaf of that frame (frame_leaf_t).
			// Directly below the frame's Composite only: an untransformed Composite in between would gate the leaf with
			// bounds it may have left, which the description does not carry for frames other than the root.
			if (P.levels_in != 0) return fail(PTX_ERR_UNSUPPORTED, "prim " + std::to_string(i) + ": a sphere, box or moved triangle below an untransformed Composite inside a transformed one");
			if (!finite || !frame_ok[P.frame]) continue;
			DevRootPrim R{};
			DevRootShade Q{};
			DevXform X{};
			R.kind_flags = P.kind;
			for (int k = 0; k < 3; ++k) {
				R.tinv[k] = P.world_inv[12 + k];
				Q.tw[k] = P.world[12 + k];
			}
			std::memcpy(R.p, P.p, sizeof R.p);
			if (P.kind == PTX_TRIANGLE)
				for (int k = 0; k < 3; ++k) {
					R.p[3 + k] = P.p[3 + k] - P.p[k];
					R.p[6 + k] = P.p[6 + k] - P.p[k];
				}
			R.id = i;
			R.frame = P.frame;
			std::memcpy(X.winv, P.world_inv, sizeof X.winv);
			std::memcpy(X.w, P.world, sizeof X.w);
			Q.n[0] = prim_info[i].x;
			Q.n[1] = prim_info[i].y;
			Q.n[2] = prim_info[i].z;
			Q.levels_in = 0;
			Q.material = P.material;
			frame_leaf_prims[P.frame].push_back(R);
			frame_leaf_shade[P.frame].push_back(Q);
			frame_leaf_xforms[P.frame].push_back(X);
			nested = true;
		}
	}
	std::vector<DevFrame> frames(D->n_frames);
	std::vector<Mat16> frame_winv(D->n_frames), frame_w(D->n_frames);
	std::vector<ptx::BvhNode> nodes;
	std::vector<float4> tris;
	for (int f = 1; f < D->n_frames; ++f) {
		const PtxFrame& F = D->frames[f];
		DevFrame& G = frames[f];
		std::memcpy(frame_winv[f].m, F.world_inv, sizeof(Mat16));
		std::memcpy(frame_w[f].m, F.world, sizeof(Mat16));
		for (int k = 0; k < 3; ++k) G.tinv[k] = F.world_inv[12 + k];
		std::memcpy(G.bmin, F.bounds_min, sizeof G.bmin);
		std::memcpy(G.bmax, F.bounds_max, sizeof G.bmax);
		G.levels_up = guarded && F.parent == 0 ? 1 : F.levels_up;  // (split root: the octant levels come with the occurrence a ray came through)
		G.parent = F.parent;
		G.plain = pure_translation(F.world) && pure_translation(F.world_inv);
		auto const& ids = frame_tris[f];
		G.has_prims = !ids.empty();
		if (guarded) {
			// the entries of the root's child this frame is, or lies in (frames come after the frame they lie in)
			if (F.parent > 0) {
				frame_occ[f] = frame_occ[F.parent];
			} else if (frame_id_hi[f] >= 0) {
				if (mesh_occ[f].empty()) return fail(PTX_ERR_INVALID, "frame " + std::to_string(f) + ": a mesh of a split root without an occurrence");
				frame_occ[f] = make_int4((int) occ.size(), (int) mesh_occ[f].size(), frame_id_lo[f], 0);
				occ.insert(occ.end(), mesh_occ[f].begin(), mesh_occ[f].end());
			}
		}
		if (ids.empty()) continue;
		std::vector<ptx::Aabb> boxes(ids.size());
		for (size_t k = 0; k < ids.size(); ++k) {
			const float* p = D->prims[ids[k]].p;
			for (int a = 0; a < 3; ++a) {
				boxes[k].lo[a] = std::min(p[a], std::min(p[3 + a], p[6 + a]));
				boxes[k].hi[a] = std::max(p[a], std::max(p[3 + a], p[6 + a]));
			}
		}
		ptx::Bvh bvh = ptx::build_bvh(boxes);
		if (bvh.max_depth > kStack) return fail(PTX_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
		sc->max_depth = std::max(sc->max_depth, bvh.max_depth);
		G.bvh_root = bvh.root_ref;
		G.node_base = (int32_t) nodes.size() * 4;
		G.tri_base = (int32_t) tris.size();
		nodes.insert(nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
		for (int32_t k : bvh.order) {
			int id = ids[k];
			const float* p = D->prims[id].p;
			float e1[3], e2[3];
			for (int a = 0; a < 3; ++a) {
				e1[a] = p[3 + a] - p[a];
				e2[a] = p[6 + a] - p[a];
			}
			float idf;
			std::memcpy(&idf, &id, 4);
			tris.push_back(make_float4(p[0], p[1], p[2], e1[0]));
			tris.push_back(make_float4(e1[1], e1[2], e2[0], e2[1]));
			tris.push_back(make_float4(e2[2], idf, 0, 0));
		}
	}
	// ---- many leaves in frame 0: a BVH of their own over the padded bounding boxes (the filter
	// boxes), the four per-leaf tables put into its leaf order.  Up to kRootBvhMin leaves the
	// wave-uniform list with its scalar loads is the faster way.
	if ((int) root_prims.size() > kRootBvhMin) {
		std::vector<ptx::Aabb> boxes(root_prims.size());
		bool finite = true;
		for (size_t k = 0; k < boxes.size(); ++k) {
			const float4 lo = root_bound[2 * k], hi = root_bound[2 * k + 1];
			boxes[k] = {{lo.x, lo.y, lo.z}, {hi.x, hi.y, hi.z}};
			finite = finite && std::isfinite(lo.x) && std::isfinite(lo.y) && std::isfinite(lo.z) && std::isfinite(hi.x) && std::isfinite(hi.y) && std::isfinite(hi.z);
		}
		if (finite) {
			// one shape per leaf: the box a ray must enter before a shape's exact test runs is then the shape's own, as in
			// the list of a small frame 0 (a leaf of four shapes - spheres, boxes and triangles mixed - ran all four
			// tests, each kind's code in turn, for every ray that touched their common box: 9 exact tests per shadow ray
			// on a room with 54 shapes in it)
			ptx::Bvh bvh = ptx::build_bvh(boxes, PTX_ROOT_BVH_LEAF);
			if (bvh.max_depth > kStack) return fail(PTX_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
			sc->max_depth = std::max(sc->max_depth, bvh.max_depth);
			std::vector<DevRootPrim> rp(root_prims.size());
			std::vector<DevRootShade> rs(root_prims.size());
			std::vector<DevXform> rx(root_prims.size());
			std::vector<float4> rb(root_bound.size());
			std::vector<int2> ro(root_occ.size());
			for (size_t k = 0; k < bvh.order.size(); ++k) {
				const int32_t from = bvh.order[k];
				if (guarded) ro[k] = root_occ[from];
				rp[k] = root_prims[from];
				rp[k].xform = (int32_t) k;
				rs[k] = root_shade[from];
				rx[k] = xforms[from];
				rb[2 * k] = root_bound[2 * from];
				rb[2 * k + 1] = root_bound[2 * from + 1];
			}
			root_prims.swap(rp);
			root_shade.swap(rs);
			xforms.swap(rx);
			root_bound.swap(rb);
			root_occ.swap(ro);
			sc->dev.root_bvh = 1;
			sc->dev.root_bvh_root = bvh.root_ref;
			sc->dev.root_node_base = (int32_t) nodes.size() * 4;
			nodes.insert(nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
		}
	}
	sc->n_nodes = (int) nodes.size();
	// the leaves of the other frames behind the root's (a hit names a leaf by its slot in these tables)
	const size_t n_root_leaves = root_prims.size();
	std::vector<int2> frame_leaves(D->n_frames, make_int2(0, 0));
	for (int f = 1; f < D->n_frames; ++f) {
		frame_leaves[f] = make_int2((int) root_prims.size(), (int) frame_leaf_prims[f].size());
		for (size_t k = 0; k < frame_leaf_prims[f].size(); ++k) {
			DevRootPrim R = frame_leaf_prims[f][k];
			R.xform = (int32_t) root_prims.size();
			root_prims.push_back(R);
			root_shade.push_back(frame_leaf_shade[f][k]);
			xforms.push_back(frame_leaf_xforms[f][k]);
		}
	}

	PtxStatus st;
	DevScene& S = sc->dev;
	if ((st = sc->upload(root_prims, &S.root_prims)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(root_shade, &S.root_shade)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	{
		const float4* rb = nullptr;
		if ((st = sc->upload(root_bound, &rb)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		std::memcpy(&S.root_bound_k, &rb, sizeof(void*));
	}
	if ((st = sc->upload(xforms, &S.xforms)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(frame_leaves, &S.frame_leaves)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(frames, &S.frames)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(frame_winv, &S.frame_winv)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(frame_w, &S.frame_w)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	{
		const ptx::BvhNode* dn = nullptr;
		if ((st = sc->upload(nodes, &dn)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		S.nodes = reinterpret_cast<const float4*>(dn);
	}
	if ((st = sc->upload(tris, &S.tris)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	if ((st = sc->upload(prim_info, &S.prim_info)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	{
		std::vector<float4> prim_mats((size_t) std::max(D->n_prims, 1) * 4);
		for (int i = 0; i < D->n_prims; ++i) std::memcpy(&prim_mats[(size_t) i * 4], &mats[(size_t) D->prims[i].material * 4], 4 * sizeof(float4));
		if ((st = sc->upload(prim_mats, &S.materials)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	}
	if ((st = sc->upload(mats, &S.mat_table)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	S.n_guards = 0;
	if (guarded) {
		if (root_prims.size() >= (size_t) 1 << kExtraShift) { ptx_scene_destroy(sc.release()); return fail(PTX_ERR_UNSUPPORTED, "a split root with 2^24 leaves or more"); }
		std::vector<float4> guards((size_t) D->n_guards * 2);
		for (int g = 0; g < D->n_guards; ++g) {
			const PtxGuard& G = D->guards[g];
			float parent_bits, extra_bits;
			const int32_t extra = G.levels - 1;
			std::memcpy(&parent_bits, &G.parent, 4);
			std::memcpy(&extra_bits, &extra, 4);
			guards[2 * g] = make_float4(G.bounds_min[0], G.bounds_min[1], G.bounds_min[2], parent_bits);
			guards[2 * g + 1] = make_float4(G.bounds_max[0], G.bounds_max[1], G.bounds_max[2], extra_bits);
		}
		if (occ.empty()) occ.push_back(make_int2(-1, 0));
		if (root_occ.empty()) root_occ.push_back(make_int2(0, 0));
		if ((st = sc->upload(guards, &S.guards)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		if ((st = sc->upload(occ, &S.occ)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		if ((st = sc->upload(root_occ, &S.root_occ)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		if ((st = sc->upload(frame_occ, &S.frame_occ)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
		S.n_guards = D->n_guards;
	}
	if ((st = sc->upload(lights, &S.lights)) != PTX_OK) { ptx_scene_destroy(sc.release()); return st; }
	// same addresses; copied as bits because the device pass types the *_k members differently
	std::memcpy(&S.root_prims_k, &S.root_prims, sizeof(void*));
	std::memcpy(&S.xforms_k, &S.xforms, sizeof(void*));
	std::memcpy(&S.frames_k, &S.frames, sizeof(void*));
	std::memcpy(&S.frame_winv_k, &S.frame_winv, sizeof(void*));
	std::memcpy(&S.lights_k, &S.lights, sizeof(void*));
	S.n_root_prims = (int) n_root_leaves;
	S.nested = nested ? 1 : 0;
	S.n_frames = D->n_frames;
	S.n_lights = D->n_lights;
	S.n_prims = D->n_prims;
	S.n_materials = D->n_materials;
	std::memcpy(S.ambient, D->ambient, sizeof S.ambient);
	std::memcpy(S.root_min, D->frames[0].bounds_min, sizeof S.root_min);
	std::memcpy(S.root_max, D->frames[0].bounds_max, sizeof S.root_max);
	// the answer to a ray with a NaN direction (nan_ray_hit): the first leaf in visiting order (= prim order) that
	// is not a sphere and was kept; a box is "hit" at infinity, a triangle at distance NaN
	S.nan_id = -1;
	S.nan_frame = S.nan_slot = S.nan_extra = 0;
	S.nan_t = 0.0f;
	auto octant_levels_above = [&](int frame) {  // split root: of the first entry of the root's child `frame` is, or lies in
		while (D->frames[frame].parent > 0) frame = D->frames[frame].parent;
		return mesh_occ[frame].empty() || mesh_occ[frame].front().x < 0 ? 0 : D->guards[mesh_occ[frame].front().x].levels - 1;
	};
	for (int i = 0; i < D->n_prims && S.nan_id < 0; ++i) {
		const PtxPrim& P = D->prims[i];
		if (P.kind == PTX_SPHERE) continue;
		if (P.frame == 0 || !(P.kind == PTX_TRIANGLE && is_identity(P.world))) {  // a leaf of the root, or a leaf kept in another frame's list
			for (size_t k = 0; k < root_prims.size(); ++k)
				if (root_prims[k].id == i) {
					S.nan_id = i;
					S.nan_slot = (int32_t) k;
					if (guarded && P.frame == 0) S.nan_extra = prim_occ[i].front().x >= 0 ? D->guards[prim_occ[i].front().x].levels - 1 : 0;
					if (guarded && P.frame > 0) S.nan_extra = octant_levels_above(P.frame);
				}
		} else if (frames[P.frame].has_prims) {
			const DevFrame& G = frames[P.frame];
			const size_t n_tri = frame_tris[P.frame].size();
			for (size_t k = 0; k < n_tri; ++k) {
				int32_t id;
				std::memcpy(&id, &tris[(size_t) G.tri_base + 3 * k + 2].y, 4);
				if (id == i) {
					S.nan_id = i;
					S.nan_frame = P.frame;
					S.nan_slot = (int32_t) k;
					if (guarded) S.nan_extra = octant_levels_above(P.frame);
				}
			}
		}
		if (S.nan_id >= 0) S.nan_t = P.kind == PTX_BOX ? std::numeric_limits<float>::infinity() : std::numeric_limits<float>::quiet_NaN();
	}
	std::memcpy(S.root_winv, D->frames[0].world_inv, sizeof S.root_winv);
	std::memcpy(S.leaf_winv, D->frames[0].world_inv, sizeof S.leaf_winv);
	for (int i = 0; i < D->n_prims; ++i)
		if (D->prims[i].frame > 0 && D->prims[i].kind == PTX_TRIANGLE && is_identity(D->prims[i].world)) {  // an untransformed mesh triangle's own inverse (the same bits for all of them)
			std::memcpy(S.leaf_winv, D->prims[i].world_inv, sizeof S.leaf_winv);
			break;
		}
	S.bounds_clean = 1;  // (frame-0 leaves and mesh frames with a non-finite number were dropped above)
	for (int k = 0; k < 3; ++k)
		if (std::isnan(S.root_min[k]) || std::isnan(S.root_max[k])) S.bounds_clean = 0;
	for (int g = 0; g < D->n_guards; ++g)
		for (int k = 0; k < 3; ++k)
			if (std::isnan(D->guards[g].bounds_min[k]) || std::isnan(D->guards[g].bounds_max[k])) S.bounds_clean = 0;

	// ---- ray queue memory (48 B per ray); ptx_render_accumulate cuts it into queues
	if (queue_bytes == 0) {
		// 64 GiB, or a third of what the device has free now if that is less.  Measured on the mesh scene at
		// 1920x1080, 1024 spp: 276.4 ms per image with 16 GiB, 250.3 with 32, 245.0 with 64, 244.1 with 96 (more memory
		// means fewer, larger launches); creating the scene takes 7 ms and destroying it 1 - 2 ms whatever the size
		// (the block is mapped, not touched, until a render writes to it), and a program that renders one image and
		// ends takes the same 0.37 s with 32 and with 64 GiB: a one-image caller loses nothing.
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		queue_bytes = std::min<uint64_t>(free_b / 3, 64ull << 30);
	}
	uint64_t cap = queue_bytes / 48;
	cap = std::min<uint64_t>(cap, 1ull << 33);  // 384 GiB: above any device; each queue is limited on its own (64 sub-queues of at most 2^25 records)
	cap = std::max<uint64_t>(cap, 1ull << 21);  // 96 MiB: enough for one queue per depth at the deepest tree allowed
	cap &= ~127ull;
	sc->work.qslots = cap;
	{
		void* p = nullptr;
		hipError_t e = take_block(device, cap * 48, &p, &sc->block_bytes);
		if (e != hipSuccess) {
			ptx_scene_destroy(sc.release());
			return fail(PTX_ERR_NOMEM, std::string("ray queue allocation failed: ") + hipGetErrorString(e));
		}
		sc->block = p;
		sc->device_bytes += cap * 48;
		sc->work.qmem = static_cast<float4*>(p);
	}
	{
		void* p = nullptr;
		if (hipMalloc(&p, count_bytes(kMaxSeg)) != hipSuccess) { ptx_scene_destroy(sc.release()); return fail(PTX_ERR_NOMEM, "counter allocation failed"); }
		sc->allocs.push_back(p);
		sc->work.counts = static_cast<unsigned*>(p);
		sc->work.max_seg = kMaxSeg;
		if (hipMalloc(&p, kCounterBytes) != hipSuccess) { ptx_scene_destroy(sc.release()); return fail(PTX_ERR_NOMEM, "counter allocation failed"); }
		sc->allocs.push_back(p);
		sc->work.counters = static_cast<Counters*>(p);
		if (hipMalloc(&p, kStatBytes) != hipSuccess) { ptx_scene_destroy(sc.release()); return fail(PTX_ERR_NOMEM, "counter allocation failed"); }
		sc->allocs.push_back(p);
		sc->work.stat = static_cast<unsigned long long*>(p);
	}
	sc->tails.assign((size_t) (kMaxBounces + 2) * kMaxSeg, 0u);
	*out = sc.release();
	return PTX_OK;
}

void ptx_jitter(uint32_t jitter_seed, uint32_t pixel, uint32_t sample, float* dx, float* dy) {
	float x = 0, y = 0;
	if (jitter_seed) jitter_offsets(jitter_seed, pixel, sample, x, y);
	if (dx) *dx = x;
	if (dy) *dy = y;
}

PtxStatus ptx_scene_set_profiling(PtxScene* scene, int enable) {
	if (!scene) return fail(PTX_ERR_INVALID, "scene is NULL");
	scene->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
	return PTX_OK;
}

void ptx_scene_destroy(PtxScene* scene) {
	if (!scene) return;
	(void) hipSetDevice(scene->device);
	for (hipEvent_t e : scene->events) (void) hipEventDestroy(e);
	if (scene->sums) (void) hipFree(scene->sums);
	if (scene->undo) (void) hipFree(scene->undo);
	if (scene->host_word) (void) hipHostFree(scene->host_word);
	if (scene->multi_accum) (void) hipFree(scene->multi_accum);
	if (scene->multi_color) (void) hipFree(scene->multi_color);
	if (scene->multi_rgb) (void) hipFree(scene->multi_rgb);
	if (scene->multi_stage) {
		if (hipSetDevice(scene->multi_stage_device) == hipSuccess) (void) hipFree(scene->multi_stage);
		(void) hipSetDevice(scene->device);
	}
	for (void* p : scene->allocs) (void) hipFree(p);
	if (scene->block) give_block(scene->device, scene->block, scene->block_bytes);
	delete scene;
}

void ptx_trim(void) {
	for (int d = 0; d < kMaxDevices; ++d) {
		KeptBlock k;
		{
			std::lock_guard<std::mutex> lock(g_kept_mutex);
			k = g_kept[d];
			g_kept[d] = KeptBlock{};
		}
		if (k.p && hipSetDevice(d) == hipSuccess) (void) hipFree(k.p);
	}
}

PtxStatus ptx_scene_info(const PtxScene* sc, int32_t* n_frames, int32_t* n_prims, int32_t* n_bvh_nodes, int32_t* max_bvh_depth,
                         uint64_t* device_bytes) {
	if (!sc) return fail(PTX_ERR_INVALID, "scene is NULL");
	if (n_frames) *n_frames = sc->n_frames;
	if (n_prims) *n_prims = sc->n_prims;
	if (n_bvh_nodes) *n_bvh_nodes = sc->n_nodes;
	if (max_bvh_depth) *max_bvh_depth = sc->max_depth;
	if (device_bytes) *device_bytes = sc->device_bytes;
	return PTX_OK;
}

}  // extern "C"

namespace {

// One render into d_accum with the work memory `M` (the scene's own block, or any other that
// holds the same parts).  Calls no allocating routine unless profiling has to create events
// (may_create_events) or the render keeps its sums in a buffer of the scene's (own_sums; see
// accumulate in ptx_kernels.hpp), which grows with the largest image asked for.
PtxStatus render_accumulate(PtxScene* sc, const WorkMem& M, const PtxRenderParams* rp, float* d_accum, hipStream_t stream, int check_overflow,
                            PtxStats* stats, bool may_create_events, bool own_sums) {
	BounceArgs A{};
	A.S = sc->dev;
	camera_basis(sc->camera, *rp, A.cam);
	root_sees_camera(sc->dev, A.cam);
	A.accum = d_accum;
	A.counters = M.counters;
	A.stat = M.stat;
	const unsigned ns = rp->sample_end - rp->sample_begin;
	A.ns = ns;
	A.inv_ns = 1.0 / ns;
	A.s0 = rp->sample_begin;
	const uint64_t total = (uint64_t) rp->width * rp->height * ns;
	const unsigned max_depth = sc->bouncing ? rp->max_ray_bounces : 0;  // no material spawns rays: depth 0 only

	HIP_TRY(hipMemsetAsync(M.counters, 0, kCounterBytes, stream));
	HIP_TRY(hipMemsetAsync(M.stat, 0, kStatBytes, stream));
	uint64_t launches = 0, chunks_redone = 0;
	const bool prof = sc->profiling > 0 && stats && may_create_events;
	const bool count_tests = sc->profiling > 1 && stats;
	// default: the staged pipeline; below some hundred thousand samples the fused one, whose one launch per depth
	// instead of three is what counts when a launch holds little work (0.53 against 0.71 ms for 96x64, 4 spp)
	const uint32_t pipeline = rp->pipeline & ~(uint32_t) (PTX_TRACE_ZERO_WEIGHT | PTX_REDUCE_RCCL);
	A.all_children = (rp->pipeline & PTX_TRACE_ZERO_WEIGHT) ? 1u : 0u;
	const bool staged = pipeline == PTX_PIPELINE_STAGED || (pipeline == PTX_PIPELINE_DEFAULT && total >= (1ull << 17)) || sc->dev.n_guards != 0 || sc->dev.nested != 0;
	sc->event_tag.clear();
	sc->event_begin.clear();
	// profiling: an event after every launch, which is the begin of the next launch as well when that
	// follows without a host-side wait in between (a stream runs its launches back to back; an event costs
	// some microseconds of the stream's time, so one per launch instead of two); tag = kind * kMaxDepthSlots + depth slot
	unsigned n_events = 0;
	bool chained = false;  // the last event recorded marks the begin of whatever is launched next
	auto tick = [&](unsigned kind, unsigned depth, bool begin) -> hipError_t {
		if (!prof) return hipSuccess;
		if (begin && chained) return hipSuccess;
		while (sc->events.size() <= n_events) {
			hipEvent_t e;
			hipError_t err = hipEventCreate(&e);
			if (err != hipSuccess) return err;
			sc->events.push_back(e);
		}
		hipError_t err = hipEventRecord(sc->events[n_events], stream);
		if (!begin) {
			sc->event_tag.push_back(kind * kMaxDepthSlots + (depth < (unsigned) kMaxDepthSlots ? depth : kMaxDepthSlots - 1));
			sc->event_begin.push_back(n_events - 1);
			chained = true;
		}
		++n_events;
		return err;
	};
	auto count_of = [&](unsigned depth) { return M.counts + (size_t) depth * M.max_seg; };
	auto shadow_count_of = [&](unsigned depth) { return M.counts + (size_t) (kMaxBounces + 2 + depth) * M.max_seg; };
	auto hit_count_of = [&](unsigned depth) { return M.counts + (size_t) (2 * (kMaxBounces + 2) + depth) * M.max_seg; };
	unsigned* const order = M.counts + (size_t) 3 * (kMaxBounces + 2) * M.max_seg;
	// one depth of the ray tree: the fused kernel, or closest hit -> shading -> shadow rays
	const bool ip = sc->int_pow;
	// the general variant of the traversal kernels: frame-0 leaves under a BVH of their own, or a split root
	// (the fused kernel has no such variant: it takes the leaves one by one and knows no octants)
	const bool rb = sc->dev.root_bvh != 0 || sc->dev.n_guards != 0 || sc->dev.nested != 0;
	static const bool use_order = !getenv("PTX_NO_ORDER");

	// Sizing (see the note on queues in ptx_kernels.hpp): what a launch takes of segment w of
	// its input lands in segment w of its outputs, at most `per_ray` records per input ray in
	// any one output queue (children: 2 with a two-child material, else 1; shadow rays: one per
	// light; hits: 1).  So a launch may take `take` = seg_cap / per_ray records of every
	// segment and nothing can overflow, whatever the materials do.
	unsigned per_ray = sc->branching ? 2u : 1u;
	if (staged) per_ray = std::max(per_ray, (unsigned) std::max(sc->dev.n_lights, 1));
	// Two ways through a chunk of primary samples.
	//   straight: every depth is launched once, back to back.  Depth d reads the queue depth d-1 wrote and that
	//     queue is dead afterwards, so two ray queues take turns and the memory goes into larger chunks.  What
	//     this cannot do is come back for the rays a launch had to leave behind: a segment that holds more than
	//     `take` records - trees of two-child materials only; a one-child material never hands on more rays
	//     than it took, and such scenes go through the whole image without the host looking at anything.  With
	//     two-child materials the kernel that sees such a segment raises Counters::remainder; the host reads
	//     that one word after every chunk, and if it is raised it puts back the chunk's pixels of the render's
	//     sums (and the work counters) as they were before the chunk and renders the chunk again the careful way.
	//   careful: one queue per depth, the counts read back after every primary chunk, remainders traced
	//     deepest depth first (below).  Also the only way where the render adds into the caller's buffer
	//     directly, and with the fused pipeline.
	const bool can_remain = sc->branching && max_depth > 0;
	// (without two-child materials the straight way needs nothing put back, hence no memory of the library's own and
	// no look at the device: with check_overflow = 0 and no statistics asked for such a render is launches and
	// memsets only, which a stream under capture accepts)
	const bool straight = max_depth > 0 && (!can_remain || (staged && own_sums));
	// the render's own sums (doubles, see accumulate): wherever the library may hold memory of its own
	const size_t n_sums = (size_t) rp->width * rp->height * 3;
	A.accum64 = nullptr;
	if (own_sums) {
		if (sc->sums_len < n_sums) {
			if (sc->sums) (void) hipFree(sc->sums);
			sc->sums = nullptr;
			sc->sums_len = 0;
			hipError_t e = hipMalloc(reinterpret_cast<void**>(&sc->sums), n_sums * sizeof(double));
			if (e != hipSuccess) return fail(PTX_ERR_NOMEM, std::string("the render's buffer of sums: ") + hipGetErrorString(e));
			sc->sums_len = n_sums;
		}
		A.accum64 = sc->sums;
		HIP_TRY(hipMemsetAsync(sc->sums, 0, n_sums * sizeof(double), stream));
	}
	unsigned nqueues = 0, n_seg = 0, seg_cap = 0, take = 0, take0 = 0;
	bool two_queues = false;  // on the straight way through a chunk (below)
	uint64_t capd = 0, chunk = 0;
	dim3 grid(1), walk_grid(1);  // a grid is n_seg waves: in workgroups of kShadeBlock (shading) or kWalkBlock (the kernels that walk)
	auto queue_at = [&](unsigned index) {  // queue `index` of the block: capd slots of 48 bytes, as three planes
		RayQueue q;
		q.a = M.qmem + (uint64_t) index * capd * 3;
		q.b = q.a + capd;
		q.c = reinterpret_cast<float2*>(q.b + capd);
		return q;
	};
	auto size_queues = [&](bool two_queues) -> PtxStatus {
		nqueues = two_queues ? std::min(max_depth, 2u) : std::max(max_depth, 1u);
		// slots of one queue: the memory holds `nqueues` ray queues (+ shadow queue and hit records
		// for the staged pipeline: 48 + 16 bytes more per slot); record indices stay below 2^31
		capd = std::min<uint64_t>(M.qslots * 48 / (48ull * nqueues + (staged ? 64 : 0)), 1ull << 30);
		// Segments = waves of every grid: kMaxSeg (four rounds of the resident workgroups) for a
		// large render; fewer for a small one, so that every wave still gets some eight 64-sample
		// steps of the first chunk and the steps of the deeper depths stay reasonably full; fewer,
		// too, when the memory is small, so that a segment still takes whole steps.
		n_seg = M.max_seg;
		while (n_seg > 64 && (total + 63) / 64 < (uint64_t) n_seg * 8) n_seg >>= 1;
		while (n_seg > 4 && capd / n_seg / per_ray < 64) n_seg >>= 1;
		seg_cap = (unsigned) (capd / n_seg) & ~63u;
		take = std::min((seg_cap / per_ray) & ~63u, (1u << 30) / n_seg);  // records of a segment per launch
		if (max_depth == 0 && !staged) take = (1u << 26) / n_seg;         // nothing is queued at all
		if (take < 64) return fail(PTX_ERR_NOMEM, "the work memory is too small for max_ray_bounces and the number of lights of this scene");
		// Straight with two-child materials: the primary launch takes half of what a launch may take, so that a
		// segment whose rays all hit such a material (its next depths then hold twice its primary samples, where
		// the image as a whole holds a fraction) still goes through every depth in one launch each.  Only trees
		// that keep doubling leave a remainder then.
		take0 = two_queues && can_remain ? std::max(64u, (take / 2) & ~63u) : take;
		chunk = (uint64_t) take0 * n_seg;  // primary samples per launch (at most 2^30)
		grid = dim3(n_seg / (kShadeBlock / 64));
		walk_grid = dim3(n_seg / (kWalkBlock / 64));
		A.seg_cap = seg_cap;
		A.n_seg = n_seg;
		if (staged) {  // shadow queue and hit records behind the ray queues
			A.shadow = queue_at(nqueues);
			A.hits = M.qmem + ((uint64_t) nqueues + 1) * capd * 3;
		}
		return PTX_OK;
	};
#ifdef PTX_TIMELINE
	// PTX_TIMELINE="kind:depth:chunk:file": the wave table of that launch (kind 0 closest hit, 1 shading, 2 shadow), written once
	static int tl_kind = -2, tl_depth = -1, tl_chunk = 0;
	static std::string tl_path;
	static bool tl_done = false;
	if (tl_kind == -2 && getenv("PTX_TIMELINE")) {
		char buf[512] = {0};
		if (sscanf(getenv("PTX_TIMELINE"), "%d:%d:%d:%500s", &tl_kind, &tl_depth, &tl_chunk, buf) == 4) tl_path = buf;
	}
	int tl_chunks_seen = -1;
	auto timeline_dump = [&](int kind, unsigned d) {
		if (d == 0 && kind == 0) ++tl_chunks_seen;
		// (kind or depth -1: every kind / every depth of the chunk, one file each)
		if (tl_done || (tl_kind >= 0 && kind != tl_kind) || (tl_depth >= 0 && (int) d != tl_depth) || tl_chunks_seen != tl_chunk) return;
		if (tl_kind >= 0 && tl_depth >= 0) tl_done = true;
		(void) hipStreamSynchronize(stream);
		std::vector<unsigned long long> t(4 * (size_t) kMaxSeg);
		(void) hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_timeline), t.size() * sizeof(unsigned long long));
		const std::string path = tl_path + (tl_kind < 0 || tl_depth < 0 ? ".k" + std::to_string(kind) + ".d" + std::to_string(d) : std::string());
		if (FILE* f = fopen(path.c_str(), "wb")) {
			unsigned hdr[4] = {n_seg, seg_cap, (unsigned) kind, d};
			fwrite(hdr, sizeof hdr, 1, f);
			fwrite(t.data(), sizeof(unsigned long long), 4 * (size_t) n_seg, f);
			fclose(f);
		}
	};
#define PTX_TL(kind, d) timeline_dump(kind, d)
#else
#define PTX_TL(kind, d) (void) 0
#endif
	auto launch_depth = [&](unsigned d) -> PtxStatus {
		if (!staged) {
			HIP_TRY(tick(0, d, true));
			auto kern = d == 0 ? (ip ? k_bounce<true, true> : k_bounce<true, false>) : (ip ? k_bounce<false, true> : k_bounce<false, false>);
			hipLaunchKernelGGL(kern, walk_grid, dim3(kWalkBlock), walk_lds_bytes(stack_entries(sc), false), stream, A);
			HIP_TRY(tick(0, d, false));
			launches += 1;
			return PTX_OK;
		}
		A.shadow_count = shadow_count_of(d);
		A.hit_count = hit_count_of(d);
		// The segments of the depths > 0 fullest first, four at a time, for the three kernels of each depth (hits and shadow
		// rays follow the rays).  The list is made ONCE per chunk, from the counts of depth 1, and serves the deeper depths as
		// well: a segment that is full at depth 1 - its pixels lie on glass - tends to stay so (measured, ms per image on the
		// mesh scene with a list of single segments at 57 us a time: no order 287, a list per depth 288, one list per chunk
		// 282; the same made in a stream of the library's own beside the shadow kernel of depth 0: 280, not kept - a render
		// then depends on two streams and two events being scheduled against each other, and one full test run hung in a
		// small render).  On the straight way through a chunk only: there every launch takes its segments from their first
		// record on.
		const bool ordering = use_order && two_queues && n_seg >= 1024;
		A.order = nullptr;
		if (d > 0 && ordering) {
			if (d == 1) {
				hipLaunchKernelGGL(k_order, dim3(1), dim3(kOrderBlock), 0, stream, A.in_count, n_seg, seg_cap, A.in_first, A.in_take, order);
				launches += 1;
			}
			A.order = order;
		}
		HIP_TRY(tick(0, d, true));
		{
			auto pick = [&](auto counted) {
				using TC = decltype(counted);
				return d == 0 ? (rb ? k_trace<true, true, TC> : k_trace<true, false, TC>) : (rb ? k_trace<false, true, TC> : k_trace<false, false, TC>);
			};
			auto kern = count_tests ? pick(TestCount{}) : pick(NoCount{});
			hipLaunchKernelGGL(kern, walk_grid, dim3(kWalkBlock), walk_lds_bytes(stack_entries(sc, d == 0 ? 0 : 1), d != 0), stream, A);
			PTX_TL(0, d);
		}
		HIP_TRY(tick(0, d, false));
		HIP_TRY(tick(1, d, true));
		{
			// the small tables of the scene from LDS copies when they fit
			const bool lt = A.S.n_root_prims <= kLdsRoot && A.S.n_frames <= kLdsFrames && A.S.n_materials <= kLdsMats && !A.S.nested;
			auto pick = [&](auto primary) {
				constexpr bool P = decltype(primary)::value;
				return ip ? (lt ? k_shade<P, true, true> : k_shade<P, true, false>) : (lt ? k_shade<P, false, true> : k_shade<P, false, false>);
			};
			auto kern = d == 0 ? pick(std::true_type{}) : pick(std::false_type{});
			A.stage_children = sc->branching ? 128u : 64u;
			A.stage_lights = (unsigned) std::min(std::max(sc->dev.n_lights, 0), kStageLights);
			const size_t stage_bytes = (size_t) (kShadeBlock / 64) * shade_stage_f4(A.stage_children, A.stage_lights) * sizeof(float4);
			hipLaunchKernelGGL(kern, grid, dim3(kShadeBlock), stage_bytes, stream, A);
			PTX_TL(1, d);
		}
		HIP_TRY(tick(1, d, false));
		if (sc->dev.n_lights > 0) {
			HIP_TRY(tick(2, d, true));
			auto kern = count_tests ? (rb ? k_shadow<true, TestCount> : k_shadow<false, TestCount>) : (rb ? k_shadow<true, NoCount> : k_shadow<false, NoCount>);
			hipLaunchKernelGGL(kern, walk_grid, dim3(kWalkBlock), walk_lds_bytes(stack_entries(sc, d == 0 ? 2 : 3), false), stream, A);
			PTX_TL(2, d);
			HIP_TRY(tick(2, d, false));
			launches += 1;
		}
		launches += 2;
		return PTX_OK;
	};

	// Driving the ray tree through the queues: the rays traced at depth d (1..max_depth) wait in
	// queue d-1 (careful) or in one of two queues taking turns (straight).  Every depth of a
	// primary chunk is first launched once, back to back, each wave reading its segment's count
	// on the device and taking at most `take` rays of it.  The careful way then reads the counts
	// ONCE per primary chunk, and if some segment of a depth held more than `take` rays the
	// remainder is traced, deepest depth first - the queues below a depth are free by then -
	// slice by slice with the counts read back after every launch (depth-first over the slices).
	// Every wave of a launch writes its counts, so nothing needs clearing in between.
	unsigned* tails = sc->tails.data();
	auto set_depth = [&](unsigned d, unsigned first, unsigned n_take) {
		const bool last = (d == max_depth);
		A.depth = d;
		A.last_depth = last;
		A.in_first = first;
		A.in_take = n_take;
		A.in_count = count_of(d);
		A.out_count = count_of(d + 1);
		A.flag_remainder = two_queues && can_remain ? 1u : 0u;
		if (d > 0) A.in = queue_at(two_queues ? (d - 1) & 1u : d - 1);
		A.out = queue_at(last ? 0 : (two_queues ? d & 1u : d));
	};
	auto fullest = [&](const unsigned* t) {  // records in the fullest segment
		unsigned m = 0;
		for (unsigned k = 0; k < n_seg; ++k) m = std::max(m, std::min(t[k], seg_cap));
		return m;
	};
	// records [first0, n) of every segment of queue d in slices; each slice's children go to queue
	// d+1 and are finished (recursively) before the next slice overwrites that queue
	auto run = [&](auto&& self, unsigned d, unsigned first0, unsigned n) -> PtxStatus {
		for (uint64_t first = first0; first < n; first += take) {
			const bool last = (d == max_depth);
			set_depth(d, (unsigned) first, take);
			PtxStatus ls = launch_depth(d);
			if (ls != PTX_OK) return ls;
			if (last) continue;
			HIP_TRY(hipMemcpyAsync(tails, count_of(d + 1), n_seg * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			chained = false;
			const unsigned m = fullest(tails);
			if (m) {
				PtxStatus st2 = self(self, d + 1, 0, m);
				if (st2 != PTX_OK) return st2;
			}
		}
		return PTX_OK;
	};
	// primary samples [begin, end) of this render, chunk by chunk
	auto careful_chunks = [&](uint64_t begin, uint64_t end) -> PtxStatus {
		two_queues = false;
		PtxStatus zs = size_queues(false);
		if (zs != PTX_OK) return zs;
		for (uint64_t first = begin; first < end; first += chunk) {
			A.first_pix = (unsigned) (first / ns);
			A.first_s = (unsigned) (first % ns);
			A.batch = (unsigned) std::min<uint64_t>(chunk, end - first);
			for (unsigned d = 0; d <= max_depth; ++d) {
				set_depth(d, 0, take);
				PtxStatus ls = launch_depth(d);
				if (ls != PTX_OK) return ls;
			}
			if (max_depth == 0) continue;
			HIP_TRY(hipMemcpyAsync(tails, count_of(1), (size_t) max_depth * M.max_seg * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			chained = false;
			unsigned held[kMaxBounces + 2] = {0};
			for (unsigned d = 1; d <= max_depth; ++d) held[d] = fullest(tails + (size_t) (d - 1) * M.max_seg);
			for (unsigned d = max_depth; d >= 1; --d) {
				if (held[d] <= take) continue;
				PtxStatus rst = run(run, d, take, held[d]);
				if (rst != PTX_OK) return rst;
			}
		}
		return PTX_OK;
	};
	auto straight_chunks = [&]() -> PtxStatus {
		two_queues = true;
		PtxStatus zs = size_queues(true);
		if (zs != PTX_OK) return zs;
		// With two-child materials and so little memory that a wave gets fewer than four 64-sample units of a chunk,
		// a segment's share of such materials swings too far for the halved take to hold: a quarter of the chunks
		// left rays behind (2 GiB on the mesh scene at 1920x1080: 1 057 ms per image against 845 the careful way).
		if (can_remain && take0 < 256) return careful_chunks(0, total);
		// what a chunk may have to put back: its pixels of the sums (a chunk is a run of consecutive pixels), the work counters
		const uint64_t band_pixels = std::min<uint64_t>(chunk / ns + 2, (uint64_t) rp->width * rp->height);
		if (can_remain) {
			const size_t need = (size_t) band_pixels * 3 * sizeof(double) + (stats ? kStatBytes : 0);
			if (sc->undo_bytes < need) {
				if (sc->undo) (void) hipFree(sc->undo);
				sc->undo = nullptr;
				sc->undo_bytes = 0;
				hipError_t e = hipMalloc(&sc->undo, need);
				if (e != hipSuccess) return fail(PTX_ERR_NOMEM, std::string("the copy of a chunk's pixels: ") + hipGetErrorString(e));
				sc->undo_bytes = need;
			}
			if (!sc->host_word) {
				hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&sc->host_word), 64, hipHostMallocDefault);
				if (e != hipSuccess) return fail(PTX_ERR_NOMEM, std::string("pinned word: ") + hipGetErrorString(e));
			}
		}
		unsigned char* const undo_stat = static_cast<unsigned char*>(sc->undo) + (size_t) band_pixels * 3 * sizeof(double);
		for (uint64_t first = 0; first < total; first += chunk) {
			const uint64_t batch = std::min<uint64_t>(chunk, total - first);
			const uint64_t p0 = first / ns, p1 = (first + batch - 1) / ns;  // the chunk's pixels, first and last
			if (can_remain) {
				HIP_TRY(hipMemcpyAsync(sc->undo, sc->sums + p0 * 3, (size_t) (p1 - p0 + 1) * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
				if (stats) HIP_TRY(hipMemcpyAsync(undo_stat, M.stat, kStatBytes, hipMemcpyDeviceToDevice, stream));
			}
			A.first_pix = (unsigned) p0;
			A.first_s = (unsigned) (first % ns);
			A.batch = (unsigned) batch;
			for (unsigned d = 0; d <= max_depth; ++d) {
				set_depth(d, 0, take);
				PtxStatus ls = launch_depth(d);
				if (ls != PTX_OK) return ls;
			}
			if (!can_remain) continue;
			HIP_TRY(hipMemcpyAsync(sc->host_word, &M.counters->remainder, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			chained = false;
			if (!*sc->host_word) continue;
			++chunks_redone;
			HIP_TRY(hipMemcpyAsync(sc->sums + p0 * 3, sc->undo, (size_t) (p1 - p0 + 1) * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
			if (stats) HIP_TRY(hipMemcpyAsync(M.stat, undo_stat, kStatBytes, hipMemcpyDeviceToDevice, stream));
			HIP_TRY(hipMemsetAsync(&M.counters->remainder, 0, sizeof(unsigned long long), stream));
			// (the launches of the discarded pass stay in `launches` and in the profiling events: they cost their time)
			PtxStatus cs = careful_chunks(first, first + batch);
			if (cs != PTX_OK) return cs;
			two_queues = true;
			if ((zs = size_queues(true)) != PTX_OK) return zs;
		}
		return PTX_OK;
	};
	{
		PtxStatus is = straight ? straight_chunks() : careful_chunks(0, total);
		if (is != PTX_OK) return is;
	}
	if (own_sums) {
		const unsigned fold_grid = (unsigned) std::min<size_t>((n_sums + kBlock - 1) / kBlock, 8192);
		hipLaunchKernelGGL(k_fold, dim3(fold_grid), dim3(kBlock), 0, stream, d_accum, sc->sums, n_sums);
	}
	HIP_TRY(hipGetLastError());
	if (check_overflow || stats) {
		Counters& c = sc->host_counters;
		if (stats) hipLaunchKernelGGL(k_stats_reduce, dim3(kMaxDepthSlots), dim3(kBlock), 0, stream, M.stat, M.counters);
		HIP_TRY(hipMemcpyAsync(&c, M.counters, sizeof c, hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		if (stats) {
			std::memset(stats, 0, sizeof *stats);
			stats->samples = total;
			stats->trace_rays = c.trace_rays;
			stats->shadow_rays = c.shadow_rays;
			stats->shaded_hits = c.shaded_hits;
			stats->launches = launches;
			stats->chunks_redone = chunks_redone;
			for (int k = 0; k < kMaxDepthSlots; ++k) {
				stats->per_depth_rays[k] = c.per_depth[k];
				stats->per_depth_shadow[k] = c.per_depth_shadow[k];
				stats->per_depth_hits[k] = c.per_depth_hits[k];
				for (int j = 0; j < kTestKinds; ++j) {
					stats->trace_tests[k][j] = c.tests[0][k][j];
					stats->shadow_tests[k][j] = c.tests[1][k][j];
				}
			}
			stats->tests_counted = count_tests ? 1 : 0;
			if (prof)
				for (size_t k = 0; k < sc->event_tag.size(); ++k) {
					float ms = 0;
					HIP_TRY(hipEventElapsedTime(&ms, sc->events[sc->event_begin[k]], sc->events[sc->event_begin[k] + 1]));
					unsigned kind = sc->event_tag[k] / kMaxDepthSlots, slot = sc->event_tag[k] % kMaxDepthSlots;
					stats->kernel_ms[kind][slot] += ms;
					stats->kernel_launches[kind][slot] += 1;
				}
		}
		if (c.overflow)
			return fail(PTX_ERR_OVERFLOW, "a ray queue overflowed (" + std::to_string(c.overflow) + " waves dropped rays); the image is incomplete");
	}
	return PTX_OK;
}

}  // namespace

extern "C" {

PtxStatus ptx_render_accumulate(PtxScene* sc, const PtxRenderParams* rp, float* d_accum, void* stream_, int check_overflow,
                                PtxStats* stats) {
	if (!sc || !d_accum) return fail(PTX_ERR_INVALID, "scene/d_accum is NULL");
	PtxStatus st = check_params(rp);
	if (st != PTX_OK) return st;
	HIP_TRY(hipSetDevice(sc->device));
	return render_accumulate(sc, sc->work, rp, d_accum, static_cast<hipStream_t>(stream_), check_overflow, stats, true, true);
}

PtxStatus ptx_render_accumulate_static(PtxScene* sc, const PtxRenderParams* rp, float* d_accum, void* stream_, int check_overflow,
                                       PtxStats* stats) {
	if (!sc || !d_accum) return fail(PTX_ERR_INVALID, "scene/d_accum is NULL");
	PtxStatus st = check_params(rp);
	if (st != PTX_OK) return st;
	const uint64_t samples = (uint64_t) rp->width * rp->height * (rp->sample_end - rp->sample_begin);
	if (samples > PTX_STATIC_MAX_SAMPLES)
		return fail(PTX_ERR_INVALID, "ptx_render_accumulate_static: " + std::to_string(samples) + " samples in one call, the static work memory takes at most " +
		                                 std::to_string((uint64_t) PTX_STATIC_MAX_SAMPLES) + " (render a smaller sample range per call, or use ptx_render_accumulate)");
	if (rp->max_ray_bounces > PTX_STATIC_MAX_BOUNCES)
		return fail(PTX_ERR_INVALID, "ptx_render_accumulate_static: max_ray_bounces " + std::to_string(rp->max_ray_bounces) + " exceeds the limit of " +
		                                 std::to_string((unsigned) PTX_STATIC_MAX_BOUNCES) + " of the static work memory");
	HIP_TRY(hipSetDevice(sc->device));
	unsigned char* base = nullptr;
	{
		std::lock_guard<std::mutex> lock(g_scratch_mutex);
		base = sc->device < kMaxDevices ? g_scratch_ptr[sc->device] : nullptr;
	}
	if (!base) return fail(PTX_ERR_DEVICE, "the static work memory of this device was never set up (ptx_scene_create does that)");
	WorkMem M;
	M.counters = reinterpret_cast<Counters*>(base);
	M.stat = reinterpret_cast<unsigned long long*>(base + kStaticStatOff);
	M.counts = reinterpret_cast<unsigned*>(base + kStaticCountsOff);
	M.max_seg = kStaticSeg;
	M.qmem = reinterpret_cast<float4*>(base + kStaticQueueOff);
	M.qslots = ((kStaticBytes - kStaticQueueOff) / 48) & ~127ull;
	return render_accumulate(sc, M, rp, d_accum, static_cast<hipStream_t>(stream_), check_overflow, stats, false, false);
}

PtxStatus ptx_resolve(int device, const float* d_accum, uint32_t w, uint32_t h, uint32_t spp, float* d_color, uint8_t* d_rgb8,
                      void* stream_) {
	if (!d_accum || w == 0 || h == 0 || spp == 0) return fail(PTX_ERR_INVALID, "bad resolve arguments");
	if ((uint64_t) w * h > 0x7fffffffull / 3) return fail(PTX_ERR_INVALID, "image too large");
	HIP_TRY(hipSetDevice(device));
	hipStream_t stream = static_cast<hipStream_t>(stream_);
	// framework/renderer.cpp:86: traced_color *= 1.0f / (aa_steps_ * aa_steps_)
	float inv_spp = 1.0f / (float) spp;
	unsigned n = w * h;
	unsigned grid = std::min<unsigned>((n + kBlock - 1) / kBlock, 4096u);
	hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, stream, d_accum, w, h, inv_spp, d_color, d_rgb8);
	HIP_TRY(hipGetLastError());
	return PTX_OK;
}

PtxStatus ptx_render_image(PtxScene* sc, const PtxRenderParams* rp, float* color, uint8_t* rgb8, double* seconds, PtxStats* stats) {
	if (!sc) return fail(PTX_ERR_INVALID, "scene is NULL");
	PtxStatus st = check_params(rp);
	if (st != PTX_OK) return st;
	HIP_TRY(hipSetDevice(sc->device));
	size_t npix = (size_t) rp->width * rp->height;
	float* d_accum = nullptr;
	float* d_color = nullptr;
	uint8_t* d_rgb = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto cleanup = [&]() {
		if (d_accum) (void) hipFree(d_accum);
		if (d_color) (void) hipFree(d_color);
		if (d_rgb) (void) hipFree(d_rgb);
		if (e0) (void) hipEventDestroy(e0);
		if (e1) (void) hipEventDestroy(e1);
	};
#define TRY_C(expr)                                                                                  \
	do {                                                                                             \
		hipError_t e_ = (expr);                                                                      \
		if (e_ != hipSuccess) {                                                                      \
			cleanup();                                                                               \
			return fail(PTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
		}                                                                                            \
	} while (0)
	TRY_C(hipMalloc(&d_accum, npix * 3 * sizeof(float)));
	TRY_C(hipMalloc(&d_color, npix * 3 * sizeof(float)));
	TRY_C(hipMalloc(&d_rgb, npix * 3));
	TRY_C(hipEventCreate(&e0));
	TRY_C(hipEventCreate(&e1));
	TRY_C(hipMemsetAsync(d_accum, 0, npix * 3 * sizeof(float), nullptr));
	TRY_C(hipEventRecord(e0, nullptr));
	PtxRenderParams full = *rp;
	st = ptx_render_accumulate(sc, &full, d_accum, nullptr, 1, stats);
	if (st != PTX_OK) { cleanup(); return st; }
	unsigned spp = rp->sample_end - rp->sample_begin;
	// a partial sample range is averaged over the WHOLE grid only by the caller that sums
	// the ranges; here the image is what this range alone gives
	st = ptx_resolve(sc->device, d_accum, rp->width, rp->height, spp ? spp : 1, d_color, d_rgb, nullptr);
	if (st != PTX_OK) { cleanup(); return st; }
	TRY_C(hipEventRecord(e1, nullptr));
	TRY_C(hipEventSynchronize(e1));
	if (seconds) {
		float ms = 0;
		TRY_C(hipEventElapsedTime(&ms, e0, e1));
		*seconds = ms * 1e-3;
	}
	if (color) TRY_C(hipMemcpy(color, d_color, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
	if (rgb8) TRY_C(hipMemcpy(rgb8, d_rgb, npix * 3, hipMemcpyDeviceToHost));
	cleanup();
#undef TRY_C
	return PTX_OK;
}

static PtxStatus run_intersect(PtxScene* sc, IntersectArgs& A, int64_t n, int32_t* prim, float* t, float* pos, float* nrm, float* rdir,
                               const float* o, const float* d) {
	std::vector<void*> tmp;
	auto cleanup = [&]() {
		for (void* p : tmp) (void) hipFree(p);
	};
	auto dalloc = [&](size_t bytes, void** p) -> hipError_t {
		hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
		if (e == hipSuccess) tmp.push_back(*p);
		return e;
	};
#define TRY_C(expr)                                                                         \
	do {                                                                                    \
		hipError_t e_ = (expr);                                                             \
		if (e_ != hipSuccess) {                                                             \
			cleanup();                                                                      \
			return fail(PTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
		}                                                                                   \
	} while (0)
	void *d_o = nullptr, *d_d = nullptr, *d_prim = nullptr, *d_t = nullptr, *d_pos = nullptr, *d_nrm = nullptr, *d_rd = nullptr;
	if (o) {
		TRY_C(dalloc((size_t) n * 12, &d_o));
		TRY_C(dalloc((size_t) n * 12, &d_d));
		TRY_C(hipMemcpy(d_o, o, (size_t) n * 12, hipMemcpyHostToDevice));
		TRY_C(hipMemcpy(d_d, d, (size_t) n * 12, hipMemcpyHostToDevice));
	}
	if (prim) TRY_C(dalloc((size_t) n * 4, &d_prim));
	if (t) TRY_C(dalloc((size_t) n * 4, &d_t));
	if (pos) TRY_C(dalloc((size_t) n * 12, &d_pos));
	if (nrm) TRY_C(dalloc((size_t) n * 12, &d_nrm));
	if (rdir) TRY_C(dalloc((size_t) n * 12, &d_rd));
	A.S = sc->dev;
	A.o = static_cast<const float*>(d_o);
	A.d = static_cast<const float*>(d_d);
	A.n = n;
	A.prim = static_cast<int32_t*>(d_prim);
	A.t = static_cast<float*>(d_t);
	A.pos = static_cast<float*>(d_pos);
	A.nrm = static_cast<float*>(d_nrm);
	A.rdir = static_cast<float*>(d_rd);
	unsigned grid = (unsigned) std::min<int64_t>((n + kWalkBlock - 1) / kWalkBlock, (int64_t) sc->grid * (kBlock / kWalkBlock));
	if (grid == 0) grid = 1;
	hipLaunchKernelGGL(sc->dev.root_bvh || sc->dev.n_guards || sc->dev.nested ? k_intersect<true> : k_intersect<false>, dim3(grid), dim3(kWalkBlock), walk_lds_bytes(stack_entries(sc), false), nullptr, A);
	TRY_C(hipGetLastError());
	TRY_C(hipDeviceSynchronize());
	if (prim) TRY_C(hipMemcpy(prim, d_prim, (size_t) n * 4, hipMemcpyDeviceToHost));
	if (t) TRY_C(hipMemcpy(t, d_t, (size_t) n * 4, hipMemcpyDeviceToHost));
	if (pos) TRY_C(hipMemcpy(pos, d_pos, (size_t) n * 12, hipMemcpyDeviceToHost));
	if (nrm) TRY_C(hipMemcpy(nrm, d_nrm, (size_t) n * 12, hipMemcpyDeviceToHost));
	if (rdir) TRY_C(hipMemcpy(rdir, d_rd, (size_t) n * 12, hipMemcpyDeviceToHost));
	cleanup();
#undef TRY_C
	return PTX_OK;
}

PtxStatus ptx_intersect(PtxScene* sc, int64_t n, const float* origins, const float* dirs, int32_t* prim, float* t, float* pos,
                        float* normal, float* raydir) {
	if (!sc) return fail(PTX_ERR_INVALID, "scene is NULL");
	if (n < 0 || (n > 0 && (!origins || !dirs))) return fail(PTX_ERR_INVALID, "bad ray arrays");
	if (n == 0) return PTX_OK;
	HIP_TRY(hipSetDevice(sc->device));
	IntersectArgs A{};
	return run_intersect(sc, A, n, prim, t, pos, normal, raydir, origins, dirs);
}

PtxStatus ptx_primary_hits(PtxScene* sc, const PtxRenderParams* rp, uint32_t sample, int32_t* prim, float* t) {
	if (!sc) return fail(PTX_ERR_INVALID, "scene is NULL");
	PtxStatus st = check_params(rp);
	if (st != PTX_OK) return st;
	if (sample >= rp->aa_steps * rp->aa_steps) return fail(PTX_ERR_INVALID, "sample outside the aa grid");
	HIP_TRY(hipSetDevice(sc->device));
	IntersectArgs A{};
	camera_basis(sc->camera, *rp, A.cam);
	root_sees_camera(sc->dev, A.cam);
	A.sample = sample;
	return run_intersect(sc, A, (int64_t) rp->width * rp->height, prim, t, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// PpmWriter::save (framework/ppmwriter.cpp:61-73): header, then "<value> " with a newline after
// every 19th value.  The 256 possible tokens are formed once; a 3840x2160 image (25 M values,
// about 90 MB of text) is then put together at memory speed instead of 25 M number formattings.
static std::string ppm_text(const uint8_t* rgb8, uint32_t w, uint32_t h) {
	char tok[256][4];
	uint8_t len[256];
	for (int v = 0; v < 256; ++v) len[v] = (uint8_t) std::snprintf(tok[v], sizeof tok[v], "%d", v);
	std::string s = "P3 " + std::to_string(w) + " " + std::to_string(h) + " 255 \n";
	const size_t n = (size_t) w * h * 3, head = s.size();
	s.resize(head + n * 4 + n / 19 + 8);
	char* out = &s[head];
	size_t pos = 1;
	for (size_t i = 0; i < n; ++i, ++pos) {
		const uint8_t v = rgb8[i];
		std::memcpy(out, tok[v], 3);
		out += len[v];
		*out++ = ' ';
		if (pos > 18) {
			*out++ = '\n';
			pos = 0;
		}
	}
	s.resize((size_t) (out - s.data()));
	return s;
}

PtxStatus ptx_write_ppm(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) {
	if (!path || !rgb8) return fail(PTX_ERR_INVALID, "path/rgb8 is NULL");
	FILE* f = std::fopen(path, "wb");
	if (!f) return fail(PTX_ERR_INVALID, std::string("cannot open ") + path);
	const std::string s = ppm_text(rgb8, w, h);
	bool ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
	ok = (std::fclose(f) == 0) && ok;
	return ok ? PTX_OK : fail(PTX_ERR_INVALID, std::string("short write to ") + path);
}

static PtxStatus write_blob(const char* path, const std::string& header, const void* data, size_t bytes) {
	FILE* f = std::fopen(path, "wb");
	if (!f) return fail(PTX_ERR_INVALID, std::string("cannot open ") + path);
	bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size();
	ok = ok && std::fwrite(data, 1, bytes, f) == bytes;
	ok = (std::fclose(f) == 0) && ok;
	return ok ? PTX_OK : fail(PTX_ERR_INVALID, std::string("short write to ") + path);
}

PtxStatus ptx_write_ppm_binary(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) {
	if (!path || !rgb8) return fail(PTX_ERR_INVALID, "path/rgb8 is NULL");
	return write_blob(path, "P6\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n", rgb8, (size_t) w * h * 3);
}

PtxStatus ptx_write_pfm(const char* path, const float* color, uint32_t w, uint32_t h) {
	if (!path || !color) return fail(PTX_ERR_INVALID, "path/color is NULL");
	const uint16_t probe = 1;
	if (*reinterpret_cast<const uint8_t*>(&probe) != 1) return fail(PTX_ERR_UNSUPPORTED, "big-endian host");
	return write_blob(path, "PF\n" + std::to_string(w) + " " + std::to_string(h) + "\n-1.0\n", color, (size_t) w * h * 3 * sizeof(float));
}

// ---------------------------------------------------------------------------
// image files written beside the rendering
// ---------------------------------------------------------------------------
struct PtxWriteJob {
	std::thread worker;
	void* pinned = nullptr;
	hipEvent_t copied = nullptr;
	int device = 0;
	PtxStatus status = PTX_OK;
	std::string error;
};

PtxStatus ptx_write_image_async(const char* path, PtxImageFormat format, int device, const void* d_pixels, uint32_t w, uint32_t h, void* stream_,
                                PtxWriteJob** out) {
	if (!path || !d_pixels || !out) return fail(PTX_ERR_INVALID, "path/d_pixels/job is NULL");
	*out = nullptr;
	if (w == 0 || h == 0 || (uint64_t) w * h > 0x7fffffffull / 3) return fail(PTX_ERR_INVALID, "bad image size");
	if (format != PTX_IMAGE_PPM_ASCII && format != PTX_IMAGE_PPM_BINARY && format != PTX_IMAGE_PFM) return fail(PTX_ERR_INVALID, "unknown image format");
	HIP_TRY(hipSetDevice(device));
	const size_t bytes = (size_t) w * h * 3 * (format == PTX_IMAGE_PFM ? sizeof(float) : 1);
	std::unique_ptr<PtxWriteJob> job(new PtxWriteJob);
	job->device = device;
	auto drop = [&]() {
		if (job->pinned) (void) hipHostFree(job->pinned);
		if (job->copied) (void) hipEventDestroy(job->copied);
	};
	hipError_t e = hipHostMalloc(&job->pinned, bytes, hipHostMallocDefault);
	if (e != hipSuccess) return fail(PTX_ERR_NOMEM, std::string("pinned host buffer: ") + hipGetErrorString(e));
	if ((e = hipEventCreateWithFlags(&job->copied, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipMemcpyAsync(job->pinned, d_pixels, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream_))) != hipSuccess ||
	    (e = hipEventRecord(job->copied, static_cast<hipStream_t>(stream_))) != hipSuccess) {
		drop();
		return fail(PTX_ERR_DEVICE, std::string("device-to-host copy: ") + hipGetErrorString(e));
	}
	PtxWriteJob* j = job.get();
	const std::string file(path);
	// the worker waits for the copy (not the caller), then formats and writes; the caller's stream is free as soon as
	// the copy has been queued, and the device buffer may be reused once work queued BEHIND the copy on that stream runs
	j->worker = std::thread([j, file, format, w, h]() {
		(void) hipSetDevice(j->device);
		hipError_t we = hipEventSynchronize(j->copied);
		if (we != hipSuccess) {
			j->status = PTX_ERR_DEVICE;
			j->error = std::string("waiting for the copy: ") + hipGetErrorString(we);
			return;
		}
		PtxStatus st;
		if (format == PTX_IMAGE_PPM_ASCII) st = ptx_write_ppm(file.c_str(), static_cast<const uint8_t*>(j->pinned), w, h);
		else if (format == PTX_IMAGE_PPM_BINARY) st = ptx_write_ppm_binary(file.c_str(), static_cast<const uint8_t*>(j->pinned), w, h);
		else st = ptx_write_pfm(file.c_str(), static_cast<const float*>(j->pinned), w, h);
		j->status = st;
		if (st != PTX_OK) j->error = ptx_last_error();  // (this thread's message; handed to the waiting thread below)
	});
	*out = job.release();
	return PTX_OK;
}

PtxStatus ptx_write_wait(PtxWriteJob* job) {
	if (!job) return fail(PTX_ERR_INVALID, "job is NULL");
	if (job->worker.joinable()) job->worker.join();
	(void) hipSetDevice(job->device);
	if (job->pinned) (void) hipHostFree(job->pinned);
	if (job->copied) (void) hipEventDestroy(job->copied);
	const PtxStatus st = job->status;
	const std::string msg = job->error;
	delete job;
	return st == PTX_OK ? PTX_OK : fail(st, msg);
}

// ---------------------------------------------------------------------------
// checkpoints of an accumulation buffer
// ---------------------------------------------------------------------------
namespace {
struct AccumHeader {  // 32 bytes, little endian
	char magic[8];    // "PTXACC1\0"
	uint32_t width, height;
	uint32_t sample_begin, sample_end;  // the samples of every pixel the buffer holds: [begin, end)
	uint32_t aa_steps;
	uint32_t reserved;
};
const char kAccumMagic[8] = {'P', 'T', 'X', 'A', 'C', 'C', '1', 0};
}  // namespace

PtxStatus ptx_accum_save(const char* path, int device, const float* d_accum, uint32_t w, uint32_t h, uint32_t aa_steps, uint32_t sample_begin,
                         uint32_t sample_end, void* stream_) {
	if (!path || !d_accum) return fail(PTX_ERR_INVALID, "path/d_accum is NULL");
	if (w == 0 || h == 0 || (uint64_t) w * h > 0x7fffffffull / 3) return fail(PTX_ERR_INVALID, "bad image size");
	if (sample_begin > sample_end) return fail(PTX_ERR_INVALID, "sample range is inverted");
	HIP_TRY(hipSetDevice(device));
	const size_t n = (size_t) w * h * 3;
	std::vector<float> host(n);
	HIP_TRY(hipMemcpyAsync(host.data(), d_accum, n * sizeof(float), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream_)));
	HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
	AccumHeader hd{};
	std::memcpy(hd.magic, kAccumMagic, 8);
	hd.width = w;
	hd.height = h;
	hd.sample_begin = sample_begin;
	hd.sample_end = sample_end;
	hd.aa_steps = aa_steps;
	// written under a temporary name and renamed: a checkpoint is either the old one or the new one, never half of one
	const std::string tmp = std::string(path) + ".part";
	FILE* f = std::fopen(tmp.c_str(), "wb");
	if (!f) return fail(PTX_ERR_INVALID, "cannot open " + tmp);
	bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1 && std::fwrite(host.data(), sizeof(float), n, f) == n;
	ok = (std::fclose(f) == 0) && ok;
	if (!ok || std::rename(tmp.c_str(), path) != 0) {
		std::remove(tmp.c_str());
		return fail(PTX_ERR_INVALID, std::string("short write to ") + path);
	}
	return PTX_OK;
}

PtxStatus ptx_accum_load(const char* path, int device, float* d_accum, uint32_t w, uint32_t h, uint32_t aa_steps, uint32_t* sample_begin,
                         uint32_t* sample_end, void* stream_) {
	if (!path || !d_accum) return fail(PTX_ERR_INVALID, "path/d_accum is NULL");
	FILE* f = std::fopen(path, "rb");
	if (!f) return fail(PTX_ERR_INVALID, std::string("cannot open ") + path);
	AccumHeader hd{};
	if (std::fread(&hd, sizeof hd, 1, f) != 1 || std::memcmp(hd.magic, kAccumMagic, 8) != 0) {
		std::fclose(f);
		return fail(PTX_ERR_INVALID, std::string(path) + " is not an accumulation checkpoint");
	}
	if (hd.width != w || hd.height != h || hd.aa_steps != aa_steps || hd.sample_begin > hd.sample_end) {
		std::fclose(f);
		return fail(PTX_ERR_INVALID, std::string(path) + " holds " + std::to_string(hd.width) + "x" + std::to_string(hd.height) + " aa_steps " +
		                                 std::to_string(hd.aa_steps) + ", not the image asked for");
	}
	const size_t n = (size_t) w * h * 3;
	std::vector<float> host(n);
	const bool ok = std::fread(host.data(), sizeof(float), n, f) == n && std::fgetc(f) == EOF;
	std::fclose(f);
	if (!ok) return fail(PTX_ERR_INVALID, std::string(path) + " is truncated or has trailing bytes");
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipMemcpyAsync(d_accum, host.data(), n * sizeof(float), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream_)));
	HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
	if (sample_begin) *sample_begin = hd.sample_begin;
	if (sample_end) *sample_end = hd.sample_end;
	return PTX_OK;
}

// ---------------------------------------------------------------------------
// RCCL, looked up when first asked for (the library does not link against it: a program that never sums buffers
// across GPUs, or has no RCCL, loads and runs without)
// ---------------------------------------------------------------------------
namespace {
struct Rccl {
	void* lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::string why;  // why it is not there
	bool ok = false;
};
Rccl& rccl() {
	static Rccl R;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (R.lib) break;
		}
		if (!R.lib) {
			R.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
			return;
		}
		auto sym = [&](const char* n) { return dlsym(R.lib, n); };
		R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
		R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
		R.Reduce = reinterpret_cast<decltype(R.Reduce)>(sym("ncclReduce"));
		R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
		R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
		R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
		R.ok = R.CommInitAll && R.CommDestroy && R.Reduce && R.GroupStart && R.GroupEnd && R.GetErrorString;
		if (!R.ok) R.why = "librccl.so lacks one of ncclCommInitAll / ncclCommDestroy / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
	});
	return R;
}
// communicators of this process, one set per list of devices (created once: setting one up takes longer than an image)
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;

// peer access from device `to` to the memory of device `from`: 1 enabled, 0 not possible (asked once per pair)
std::mutex g_peer_mutex;
signed char g_peer[kMaxDevices][kMaxDevices];
bool g_peer_init = false;
bool peer_readable(int to, int from) {  // `to` is the current device
	if (to == from) return true;
	if (to < 0 || from < 0 || to >= kMaxDevices || from >= kMaxDevices) return false;
	std::lock_guard<std::mutex> lock(g_peer_mutex);
	if (!g_peer_init) {
		std::memset(g_peer, -1, sizeof g_peer);
		g_peer_init = true;
	}
	if (g_peer[to][from] < 0) {
		int can = 0;
		bool ok = hipDeviceCanAccessPeer(&can, to, from) == hipSuccess && can;
		if (ok) {
			const hipError_t e = hipDeviceEnablePeerAccess(from, 0);
			ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
			(void) hipGetLastError();
		}
		g_peer[to][from] = ok ? 1 : 0;
	}
	return g_peer[to][from] == 1;
}
}  // namespace

PtxStatus ptx_reduce_accum(void* comm, const float* d_send, float* d_recv, uint64_t n_floats, int root, void* stream_) {
	if (!comm || !d_send) return fail(PTX_ERR_INVALID, "comm / d_send is NULL");
	Rccl& R = rccl();
	if (!R.ok) return fail(PTX_ERR_UNSUPPORTED, "RCCL is not available: " + R.why);
	const ncclResult_t r = R.Reduce(d_send, d_recv ? d_recv : const_cast<float*>(d_send), (size_t) n_floats, ncclFloat, ncclSum, root, static_cast<ncclComm_t>(comm),
	                                static_cast<hipStream_t>(stream_));
	if (r != ncclSuccess) return fail(PTX_ERR_DEVICE, std::string("ncclReduce: ") + R.GetErrorString(r));
	return PTX_OK;
}

// ---------------------------------------------------------------------------
// one image on several GPUs
// ---------------------------------------------------------------------------
namespace {
// a buffer of a scene that ptx_render_image_multi keeps from call to call, grown when a larger image is asked for
hipError_t grow(void** p, size_t* have, size_t need) {
	if (*have >= need) return hipSuccess;
	if (*p) (void) hipFree(*p);
	*p = nullptr;
	*have = 0;
	const hipError_t e = hipMalloc(p, need);
	if (e == hipSuccess) *have = need;
	return e;
}
}  // namespace

PtxStatus ptx_render_image_multi(PtxScene* const* scenes, int32_t n_scenes, const PtxRenderParams* rp, float* color, uint8_t* rgb8, double* seconds,
                                 PtxStats* stats) {
	if (!scenes || n_scenes < 1) return fail(PTX_ERR_INVALID, "no scenes");
	if (n_scenes > kMaxPeers + 1) return fail(PTX_ERR_INVALID, "more than " + std::to_string(kMaxPeers + 1) + " scenes");
	bool distinct = true;
	for (int i = 0; i < n_scenes; ++i) {
		if (!scenes[i]) return fail(PTX_ERR_INVALID, "scene " + std::to_string(i) + " is NULL");
		for (int j = 0; j < i; ++j) {
			if (scenes[j] == scenes[i]) return fail(PTX_ERR_INVALID, "the same scene object twice: every worker needs its own (ptx_scene_create once per GPU)");
			if (scenes[j]->device == scenes[i]->device) distinct = false;
		}
	}
	PtxStatus st = check_params(rp);
	if (st != PTX_OK) return st;
	const auto t0 = std::chrono::steady_clock::now();
	const size_t n = (size_t) rp->width * rp->height * 3;
	const uint32_t ns = rp->sample_end - rp->sample_begin;
	struct Worker {
		PtxStatus status = PTX_OK;
		std::string error;
	};
	std::vector<Worker> work((size_t) n_scenes);
	// the reference's fan-out over CPU threads (framework/renderer.cpp:39-53), over GPUs: worker r
	// renders all pixels for the r-th slice of the per-pixel samples into a buffer on its own GPU (the
	// scene's own, kept from call to call)
	auto run = [&](int r) {
		Worker& w = work[(size_t) r];
		PtxScene* sc = scenes[r];
		auto bad = [&](PtxStatus s, std::string m) {
			w.status = s;
			w.error = std::move(m);
		};
		hipError_t e = hipSetDevice(sc->device);
		if (e == hipSuccess) e = grow(reinterpret_cast<void**>(&sc->multi_accum), &sc->multi_accum_bytes, n * sizeof(float));
		if (e == hipSuccess) e = hipMemsetAsync(sc->multi_accum, 0, n * sizeof(float), nullptr);
		if (e != hipSuccess) return bad(PTX_ERR_DEVICE, std::string("worker set-up: ") + hipGetErrorString(e));
		PtxRenderParams part = *rp;
		part.pipeline = rp->pipeline & ~(uint32_t) PTX_REDUCE_RCCL;
		part.sample_begin = rp->sample_begin + (uint32_t) ((uint64_t) ns * (uint64_t) r / (uint64_t) n_scenes);
		part.sample_end = rp->sample_begin + (uint32_t) ((uint64_t) ns * (uint64_t) (r + 1) / (uint64_t) n_scenes);
		PtxStats* ps = stats ? stats + r : nullptr;
		if (ps) std::memset(ps, 0, sizeof *ps);
		if (part.sample_begin < part.sample_end) {
			PtxStatus rs = render_accumulate(sc, sc->work, &part, sc->multi_accum, nullptr, 1, ps, true, true);
			if (rs != PTX_OK) return bad(rs, ptx_last_error());
		}
		e = hipStreamSynchronize(nullptr);
		if (e != hipSuccess) bad(PTX_ERR_DEVICE, std::string("worker: ") + hipGetErrorString(e));
	};
	{
		std::vector<std::thread> threads;
		for (int r = 1; r < n_scenes; ++r) threads.emplace_back(run, r);
		run(0);
		for (auto& t : threads) t.join();
	}
	for (int r = 0; r < n_scenes; ++r)
		if (work[(size_t) r].status != PTX_OK) return fail(work[(size_t) r].status, "GPU worker " + std::to_string(r) + ": " + work[(size_t) r].error);
	PtxScene* const s0 = scenes[0];
	const int dev0 = s0->device;
	HIP_TRY(hipSetDevice(dev0));
	// ---- the sum of the buffers on the first scene's GPU
	bool summed = n_scenes == 1;
	if (!summed && (rp->pipeline & PTX_REDUCE_RCCL)) {
		// one RCCL reduce (sum, float) onto the first GPU: one communicator per GPU of this process, the calls of all GPUs in one group
		if (!distinct) return fail(PTX_ERR_INVALID, "PTX_REDUCE_RCCL needs every scene on a GPU of its own (RCCL refuses two ranks on one device)");
		Rccl& R = rccl();
		if (!R.ok) return fail(PTX_ERR_UNSUPPORTED, "RCCL is not available: " + R.why);
		std::vector<int> devs;
		for (int r = 0; r < n_scenes; ++r) devs.push_back(scenes[r]->device);
		std::vector<ncclComm_t> comms;
		{
			std::lock_guard<std::mutex> lock(g_comm_mutex);
			auto it = g_comms.find(devs);
			if (it == g_comms.end()) {
				std::vector<ncclComm_t> fresh((size_t) n_scenes, nullptr);
				const ncclResult_t r = R.CommInitAll(fresh.data(), n_scenes, devs.data());
				if (r != ncclSuccess) return fail(PTX_ERR_DEVICE, std::string("ncclCommInitAll: ") + R.GetErrorString(r));
				it = g_comms.emplace(devs, fresh).first;
			}
			comms = it->second;
		}
		ncclResult_t r = R.GroupStart();
		for (int k = 0; k < n_scenes && r == ncclSuccess; ++k) {
			HIP_TRY(hipSetDevice(scenes[k]->device));
			r = R.Reduce(scenes[k]->multi_accum, scenes[k]->multi_accum, n, ncclFloat, ncclSum, 0, comms[(size_t) k], nullptr);
		}
		const ncclResult_t ge = R.GroupEnd();
		if (r == ncclSuccess) r = ge;
		if (r != ncclSuccess) return fail(PTX_ERR_DEVICE, std::string("ncclReduce: ") + R.GetErrorString(r));
		for (int k = 0; k < n_scenes; ++k) {
			HIP_TRY(hipSetDevice(scenes[k]->device));
			HIP_TRY(hipStreamSynchronize(nullptr));
		}
		HIP_TRY(hipSetDevice(dev0));
		summed = true;
	}
	if (!summed) {
		// One kernel on the first GPU reads every other GPU's buffer where it lies (xGMI is point to point: the reads
		// of different peers travel over different links, side by side) and adds them in worker order.  A peer whose
		// memory this GPU cannot map is copied into a staging buffer here first.
		PeerList L{};
		for (int r = 1; r < n_scenes; ++r) {
			PtxScene* sr = scenes[r];
			const float* src = sr->multi_accum;
			if (!peer_readable(dev0, sr->device)) {
				if (sr->multi_stage && sr->multi_stage_device != dev0) {  // (staged for another GPU before)
					HIP_TRY(hipSetDevice(sr->multi_stage_device));
					(void) hipFree(sr->multi_stage);
					sr->multi_stage = nullptr;
					sr->multi_stage_bytes = 0;
					HIP_TRY(hipSetDevice(dev0));
				}
				HIP_TRY(grow(reinterpret_cast<void**>(&sr->multi_stage), &sr->multi_stage_bytes, n * sizeof(float)));
				sr->multi_stage_device = dev0;
				HIP_TRY(hipMemcpyPeerAsync(sr->multi_stage, dev0, sr->multi_accum, sr->device, n * sizeof(float), nullptr));
				src = sr->multi_stage;
			}
			L.p[L.n++] = src;
		}
		const unsigned grid = (unsigned) std::min<size_t>((n / 4 + kBlock - 1) / kBlock + 1, 8192);
		hipLaunchKernelGGL(k_add_peers, dim3(grid), dim3(kBlock), 0, nullptr, s0->multi_accum, L, n);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(grow(reinterpret_cast<void**>(&s0->multi_color), &s0->multi_color_bytes, n * sizeof(float)));
	HIP_TRY(grow(reinterpret_cast<void**>(&s0->multi_rgb), &s0->multi_rgb_bytes, n));
	st = ptx_resolve(dev0, s0->multi_accum, rp->width, rp->height, ns ? ns : 1, s0->multi_color, s0->multi_rgb, nullptr);
	if (st != PTX_OK) return st;
	if (color) HIP_TRY(hipMemcpy(color, s0->multi_color, n * sizeof(float), hipMemcpyDeviceToHost));
	if (rgb8) HIP_TRY(hipMemcpy(rgb8, s0->multi_rgb, n, hipMemcpyDeviceToHost));
	HIP_TRY(hipDeviceSynchronize());
	if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return PTX_OK;
}

}  // extern "C"
