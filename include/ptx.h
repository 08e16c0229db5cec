/*
 * ptx.h - C ABI of the MI355X (gfx950) render path.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or torch types.
 * It replaces the body of the reference's Renderer::render (framework/renderer.cpp:25-59)
 * and everything that call reaches on the CPU:
 *     Renderer::thread_function / trace / shade / phong_color / reflection / refraction
 *         (framework/renderer.cpp:61-291)
 *     Composite::intersect            (framework/composite.cpp:75-103)
 *     Sphere/Box/Triangle::intersect  (framework/sphere.cpp:35-52, box.cpp:89-165,
 *                                      triangle.cpp:56-88)
 * The reference has no FFI layer of its own; the seam is the C++ class Renderer
 * (framework/renderer.hpp:22-62).  A host-side Renderer with that class's signatures
 * flattens its Scene (framework/scene.hpp:13-21) into a PtxSceneDesc and calls the
 * functions below (path-tracer_amd/csrc/host/renderer.cpp and flatten.cpp are a working binding).
 *
 * Conventions
 *   - every function returns a PtxStatus; ptx_last_error() gives the message of the
 *     calling thread's last failure.  Nothing throws across this boundary.
 *   - matrices are 16 floats, column-major (element [col*4+row]), exactly the memory of
 *     a glm::mat4 (Shape::world_transform_ / world_transform_inv_, framework/shape.hpp:39-40).
 *   - pointers named d_* are DEVICE pointers (HBM); all others are host pointers.
 *   - `stream` arguments are a hipStream_t passed as void* (NULL = the default stream).
 *   - the library is built only for gfx950; with no such GPU every call that needs the
 *     device fails with PTX_ERR_DEVICE (there is no CPU fallback).
 */
#ifndef PTX_H
#define PTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtxStatus {
	PTX_OK = 0,
	PTX_ERR_INVALID = 1,     /* bad argument / malformed scene description            */
	PTX_ERR_UNSUPPORTED = 2, /* scene shape this library does not flatten (a sphere, box or moved triangle below an untransformed Composite inside a transformed one; transformed Composites more than 32 deep) */
	PTX_ERR_DEVICE = 3,      /* HIP call failed, or no gfx950 device                  */
	PTX_ERR_OVERFLOW = 4,    /* a ray queue filled up: result discarded, see message  */
	PTX_ERR_NOMEM = 5
} PtxStatus;

/* Material (framework/material.hpp:10-27) */
typedef struct PtxMaterial {
	float ka[3], kd[3], ks[3];
	float m, glossy, opacity, ior;
} PtxMaterial;

/* PointLight (framework/pointLight.hpp:6-12): intensity = color * brightness */
typedef struct PtxLight {
	float position[3];
	float intensity[3];
} PtxLight;

/* Camera (framework/camera.hpp:8-14) */
typedef struct PtxCamera {
	float fov_x;
	float position[3], direction[3], up[3];
} PtxCamera;

/*
 * A frame is a Composite that changes coordinates and filters rays with its bounds_ box
 * (framework/composite.cpp:75-103): frame 0 is scene.root, every other frame is a
 * transformed Composite (a loaded OBJ mesh; a sub-object of one that a caller moved with
 * find_child(..)->translate / rotate / scale, composite.cpp:112-114, :168-181).  Octants and OBJ
 * sub-objects with the identity transform are NOT frames; they only show up as `levels`.
 * A frame may lie inside another frame (a transformed Composite below a transformed Composite),
 * at most 32 deep; Composite::intersect then tests one bounds_ box and changes coordinates once
 * per level, and so does the library.  Frames are listed depth first: a frame's parent comes
 * before it.
 *   bounds_min/max  the frame's bounds_ box, in the space of the frame it lies in (as the
 *                   reference leaves it: a parent's box is not rebuilt when a child moves)
 *   parent          the frame this one lies in (0 = directly under the root or its octants)
 *   levels_up       parent 0: Composites between this frame and the root, the root included,
 *                   the frame excluded; parent > 0: Composites strictly between this frame and
 *                   its parent frame's Composite (each one re-normalises the normal, :100)
 */
typedef struct PtxFrame {
	float world[16];
	float world_inv[16];
	float bounds_min[3], bounds_max[3];
	int32_t parent;    /* -1 for frame 0 */
	int32_t levels_up; /* 0 for frame 0 */
} PtxFrame;

typedef enum PtxPrimKind { PTX_SPHERE = 0, PTX_BOX = 1, PTX_TRIANGLE = 2 } PtxPrimKind;

/*
 * One leaf Shape.  The array order IS the tie-break order: element i is the i-th leaf met
 * when walking the post-build_octree Composite tree depth first with children in
 * std::map<std::string,...> order, duplicates (leaves that sit in several octants)
 * listed once at their first encounter (framework/composite.hpp:35, composite.cpp:87-96).
 *   p          sphere: center xyz, radius        (framework/sphere.hpp:22-23)
 *              box: min xyz, max xyz             (framework/box.hpp:27-28)
 *              triangle: v0, v1, v2              (framework/triangle.hpp:32-34)
 *   normal     triangle only: n_                 (framework/triangle.hpp:35)
 *   world/inv  the leaf's own transforms
 *   frame      index of the frame the leaf lives in
 *   levels_in  Composites between the leaf and its frame's own transform, i.e. the
 *              octants/sub-objects above the leaf inside the frame; for frame 0 the
 *              root itself is included (so levels_in >= 1 there)
 * Inside a frame other than the root the usual leaf is an untransformed triangle of a loaded
 * mesh (all of a frame's go into one BVH).  A sphere, a box or a triangle with a transform of
 * its own may live there too - a Composite a caller filled with shapes and moved as a whole, a
 * face of a mesh part moved on its own - if it lies directly below the frame's Composite
 * (levels_in == 0).  An UNtransformed Composite that holds such a leaf (a part of a loaded mesh
 * one face of which was moved, an octant of such a part) gates the leaf with the bounds_ it had
 * before the leaf moved (nothing rebuilds them): the hand-over lists such a Composite as a
 * frame of its own, with its identity matrices, so that its bounds test is kept - the working
 * flatten does (host/flatten.cpp, holds_other_leaves) - and PTX_ERR_UNSUPPORTED is what a
 * description gets that leaves it out (levels_in != 0 on such a leaf).
 */
typedef struct PtxPrim {
	int32_t kind;
	int32_t material;
	int32_t frame;
	int32_t levels_in;
	float p[9];
	float normal[3];
	float world[16];
	float world_inv[16];
} PtxPrim;

/*
 * Octants of the root.  A root with more than 64 children is split by build_octree
 * (framework/composite.cpp:125-160) into octant Composites, recursively; the children - leaves
 * and meshes - then sit in octants, some in several, and Composite::intersect (:75-103) only
 * reaches a child through an octant whose bounds_ box, and that of every octant above it, the ray
 * enters.  Those boxes come from Shape::min/max, which ignore a sphere's transform
 * (sphere.cpp:22-28) and use two corners of a box (box.cpp:42-66), so a child can stick out of
 * the octant it was filed under: what the reference reports then depends on the octants, and a
 * scene description that is to be answered exactly as the reference does lists them:
 *   guards       one per octant Composite: its bounds_ box, the octant above it, and the number
 *                of Composites from it up to the root (both included), which is how often a
 *                normal found through it is re-normalised (:100)
 *   occurrences  every child entry met when walking the root's octants depth first, children
 *                in std::map order - the order Composite::intersect visits them, ties going to
 *                the first: the octant the entry sits in and the leaf (index into prims) or mesh
 *                (frame index) it names.  A child of several octants has several occurrences.
 * With n_guards == 0 (a root of at most 64 children: every scene shipped with the reference)
 * both lists are ignored.  With guards, every frame-0 leaf and every frame that lies directly in
 * the root (parent == 0) needs at least one occurrence, and levels_in of frame-0 leaves / levels_up
 * of those frames are taken from the guards.  A frame inside another frame has no occurrence of its
 * own: rays reach it through those of the frame at the top of its chain.
 */
typedef struct PtxGuard {
	float bounds_min[3], bounds_max[3];
	int32_t parent; /* index of the octant above, -1: a child of the root itself */
	int32_t levels; /* 2 for an octant of the root, 3 for an octant of such an octant, ... */
} PtxGuard;

typedef struct PtxOccurrence {
	int32_t guard; /* index into guards; -1: directly under the root */
	int32_t prim;  /* a leaf: its index in prims; -1 for a mesh */
	int32_t frame; /* a mesh: its frame (>= 1, one with parent == 0); 0 for a leaf */
	int32_t reserved;
} PtxOccurrence;

typedef struct PtxSceneDesc {
	const PtxFrame* frames;
	int32_t n_frames;
	const PtxPrim* prims;
	int32_t n_prims;
	const PtxMaterial* materials;
	int32_t n_materials;
	const PtxLight* lights;
	int32_t n_lights;
	float ambient[3]; /* Light::intensity of scene.ambient (framework/light.hpp:7-12) */
	PtxCamera camera;
	const PtxGuard* guards; /* octants of the root, see above; may be NULL with n_guards == 0 */
	int32_t n_guards;
	const PtxOccurrence* occurrences;
	int32_t n_occurrences;
} PtxSceneDesc;

typedef struct PtxScene PtxScene; /* opaque: device-resident scene + work queues */

/*
 * What Renderer's constructor and render() fix for one image
 * (framework/renderer.cpp:15-21, :74-86).  A sample is one (aax, aay) grid position of a
 * pixel, numbered s = aax * aa_steps + aay; [sample_begin, sample_end) selects the part
 * of the per-pixel sample grid this call renders (multi-GPU: one range per GPU).
 */
typedef struct PtxRenderParams {
	uint32_t width, height;
	uint32_t aa_steps;
	uint32_t max_ray_bounces;
	uint32_t sample_begin, sample_end; /* 0, aa_steps*aa_steps for the whole image */
	uint32_t pipeline;                 /* PtxPipeline; 0 = the library's default */
	uint32_t jitter_seed;              /* 0 = the reference's regular grid (sample at the corner of its sub-pixel cell);
	                                      otherwise each sample moves to a pseudo-random point INSIDE its cell, drawn from a
	                                      counter-based generator keyed by (seed, pixel, sample) - see ptx_jitter().  No
	                                      counterpart in the reference, which has no random numbers at all. */
} PtxRenderParams;

/* The offsets (in units of one sub-pixel cell, each in [0, 1)) jitter_seed != 0 adds to sample
 * `sample` of pixel `pixel`; the same function the kernels evaluate.  Stateless: any GPU can
 * produce any sample's offsets, so a sample range renders the same whichever rank takes it. */
void ptx_jitter(uint32_t jitter_seed, uint32_t pixel, uint32_t sample, float* dx, float* dy);

/*
 * How one depth of the ray tree is run (same rays, same arithmetic per ray either way):
 *   FUSED   one kernel per depth: closest hit + shading + inline shadow walks + emit
 *   STAGED  three kernels per depth: closest hit -> hit records; shading -> child rays and
 *           shadow-ray records; shadow walk -> accumulate.  Leaner kernels (more waves per
 *           SIMD) at the price of hit records and shadow rays going through HBM.
 *   DEFAULT staged; fused for calls of fewer than 2^17 samples (previews, tiles), where the
 *           number of launches decides.
 * A scene with a split root (PtxSceneDesc::n_guards > 0) or with frames inside frames always takes
 * the staged kernels, whatever is asked for: the fused kernel knows neither.
 */
typedef enum PtxPipeline { PTX_PIPELINE_DEFAULT = 0, PTX_PIPELINE_FUSED = 1, PTX_PIPELINE_STAGED = 2 } PtxPipeline;

/*
 * Flag, OR-ed into PtxRenderParams::pipeline.  A reflected or refracted ray whose weight - the product of the
 * factors the reference will multiply its colour by on the way up (renderer.cpp:120-133, :194-225) - is exactly
 * zero in all three channels cannot change the pixel, and by default is not traced.  The reference traces it
 * and multiplies the result by that zero; the two differ only if the result is not finite (0 * NaN = NaN: a box
 * hit that matches no face reports a NaN normal, box.cpp:167-182).  With this flag every child is traced, and
 * the ray counts equal the reference's as well.
 */
#define PTX_TRACE_ZERO_WEIGHT 0x100u
/* or-ed into `pipeline` for ptx_render_image_multi: sum the per-GPU buffers with one RCCL reduce
 * (ncclSum, ncclFloat, onto the first GPU) instead of the first GPU reading its peers' memory */
#define PTX_REDUCE_RCCL 0x200u

/* Work done by the last ptx_render_accumulate on this scene (whole call). */
typedef struct PtxStats {
	uint64_t samples;      /* primary rays                                           */
	uint64_t trace_rays;   /* closest-hit queries (primary + reflection + refraction) */
	uint64_t shadow_rays;  /* any-hit queries                                        */
	uint64_t shaded_hits;  /* hits that were shaded                                  */
	uint64_t launches;     /* kernel launches                                        */
	uint64_t per_depth_rays[16];    /* closest-hit queries per ray-tree depth          */
	uint64_t per_depth_shadow[16];  /* any-hit queries issued at that depth            */
	uint64_t per_depth_hits[16];    /* shaded hits at that depth                       */
	/* filled only while profiling is on (ptx_scene_set_profiling): device time per kernel
	 * kind and depth, from HIP events recorded around every launch on the caller's
	 * stream, and the number of launches timed */
	/* [kind][depth]: kind 0 = the fused kernel or the closest-hit kernel, 1 = shading kernel,
	 * 2 = shadow kernel (kinds 1 and 2 only in the staged pipeline) */
	double kernel_ms[3][16];
	uint64_t kernel_launches[3][16];
	/* filled only at profiling level 2 (tests_counted = 1): what the closest-hit kernel and the
	 * shadow kernel tested, per ray-tree depth, counted per ray (lane) that ran the test:
	 * [depth][0] slab tests (bounding boxes: root and mesh bounds_, the filter in front of a
	 * root shape, 2 per BVH node visited)  [1] triangle tests  [2] sphere tests  [3] box tests
	 * [4] ray transforms into a shape's or mesh's own space.  These are the Box / Triangle /
	 * Sphere::intersect and transform_ray calls of the reference (framework/composite.cpp:78-96). */
	uint64_t trace_tests[16][5];
	uint64_t shadow_tests[16][5];
	uint64_t tests_counted;
	/* chunks of primary samples that left rays behind on the straight way through (trees of two-child
	 * materials deeper than the queues' headroom) and were rendered again the careful way */
	uint64_t chunks_redone;
} PtxStats;

const char* ptx_last_error(void);
const char* ptx_version(void);

/* Number of gfx950 devices visible (0 if none / no HIP runtime). Never fails. */
int ptx_device_count(void);

/*
 * Validates and uploads a scene to `device`, builds the per-frame BVHs and allocates the
 * ray queues.  queue_bytes = HBM budget for the ray queues (0 = 64 GiB, or a third of the
 * memory the device has free at this moment if that is less; at least 96 MiB is taken; more
 * memory means fewer, larger kernel launches, never different results: 64 GiB is where the
 * gain ends, and the block costs the same few milliseconds to create and destroy at any size).
 * Replaces the per-thread Scene copies of framework/renderer.cpp:48.
 * A PtxScene owns its queues and counters: use it from one thread (one render) at a time;
 * different scenes are independent.  ptx_last_error() is per thread.
 */
PtxStatus ptx_scene_create(const PtxSceneDesc* desc, int device, uint64_t queue_bytes, PtxScene** out);
/* Releases the scene.  Its queue block is not returned to the device at once: the library keeps the largest
 * block destroyed on each device for the next scene created there (a caller in the place of Renderer::render,
 * framework/renderer.cpp:25-59, creates and destroys one scene per image, and releasing and mapping tens of
 * GiB per image costs more than the image).  ptx_trim() returns the kept blocks. */
void ptx_scene_destroy(PtxScene* scene);
void ptx_trim(void);

/*
 * Adds, for every pixel, the sum over the selected samples of trace(primary ray)
 * (framework/renderer.cpp:74-85) to d_accum (w*h*3 floats, row-major, y up, NOT cleared
 * here).  All device work goes to `stream`.  The sums of one call are formed in doubles in a
 * buffer of the scene's own (allocated at the first call, grown with the image size) and added to
 * d_accum once per pixel at the end of the call.  Host waits on the stream: none while rendering
 * when no material of the scene spawns two child rays; with such materials one per chunk of
 * primary samples (one word is read back: whether a launch had to leave rays behind, in which
 * case that chunk is rendered again with one queue per depth; a chunk is as many samples as the
 * queue memory is certain to hold the ray tree of); and once at the end for the statistics when
 * `stats` is non-NULL or check_overflow is non-zero.
 * Launches are sized so that no queue can fill up whatever the materials do; the kernels still
 * flag a record that would not fit, and with check_overflow != 0 (or stats) the call reads that
 * flag at its end and returns PTX_ERR_OVERFLOW if it was ever raised (check_overflow = 0 and
 * stats = NULL skips that read and the wait it needs).
 */
PtxStatus ptx_render_accumulate(PtxScene* scene, const PtxRenderParams* params, float* d_accum, void* stream,
                                int check_overflow, PtxStats* stats);

/*
 * ptx_render_accumulate for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code under HIP graph capture - see the host waits below).  No
 * counterpart in the reference.  Same arguments, same kernels, same results (the sums go straight
 * into d_accum as float atomics: there is no buffer of doubles to keep them in); the queues and counters live in a
 * 128 MiB array inside the library's code object (every GPU that runs the library holds its own
 * copy until the process ends; ptx_scene_create looks its address up once per device) instead of
 * the scene's own block, and the call allocates and frees nothing: no hipMalloc / hipFree, no
 * event creation (profiling set with ptx_scene_set_profiling is ignored here: the kernel_ms fields
 * of `stats` stay zero).
 * Limits: at most PTX_STATIC_MAX_SAMPLES samples (width * height * (sample_end - sample_begin))
 * and max_ray_bounces <= PTX_STATIC_MAX_BOUNCES per call, else PTX_ERR_INVALID; PTX_ERR_NOMEM if
 * the scene's lights and bounces still do not fit.  Larger images go through in several calls with
 * consecutive sample ranges (the sums add up in d_accum).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two threads, two
 * streams, two scenes) race; run them one after the other.  What must start at zero is cleared
 * by the call itself.
 * Host waits: with check_overflow = 0 and stats = NULL the call never waits on the stream when
 * no material of the scene spawns two child rays - it then consists of memsets and kernel
 * launches only and may be issued on a stream under capture (hipStreamBeginCapture); a scene with
 * two-child materials makes it read the queue counters back once per chunk of primary samples,
 * which a capturing stream refuses.
 */
#define PTX_STATIC_MAX_SAMPLES (1u << 24)
#define PTX_STATIC_MAX_BOUNCES 16u
PtxStatus ptx_render_accumulate_static(PtxScene* scene, const PtxRenderParams* params, float* d_accum, void* stream,
                                       int check_overflow, PtxStats* stats);

/*
 * accum / spp -> tone map c/(c+1) (framework/renderer.cpp:86-89, :286-291) -> d_color
 * (w*h*3 floats, the layout of Renderer::color_buffer()) and/or d_rgb8 (w*h*3 bytes, rows
 * flipped and quantised as PpmWriter::write does, framework/ppmwriter.cpp:36-43).
 * Either output may be NULL.
 */
PtxStatus ptx_resolve(int device, const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, float* d_color,
                      uint8_t* d_rgb8, void* stream);

/*
 * Whole Renderer::render in one blocking call, host buffers out: color (w*h*3 floats)
 * and/or rgb8 (w*h*3 bytes) may be NULL.  seconds (optional) = device time of the
 * accumulate + resolve part.
 */
PtxStatus ptx_render_image(PtxScene* scene, const PtxRenderParams* params, float* color, uint8_t* rgb8,
                           double* seconds, PtxStats* stats);

/*
 * Renderer::render's fan-out over hardware threads (framework/renderer.cpp:39-53), over GPUs:
 * one image rendered by n_scenes GPUs of this process.  scenes[r] is the same scene created on
 * GPU r (ptx_scene_create once per device; two scenes on one device also work, for rehearsals
 * on a one-GPU machine).  The call starts one host thread per scene; worker r renders ALL pixels
 * for the r-th of n_scenes contiguous slices of [sample_begin, sample_end) into a buffer on its
 * own GPU; the buffers are then summed on scenes[0]'s GPU (every peer's buffer comes over its
 * own xGMI link and is added by a kernel), resolved there, and copied to the host arrays like
 * ptx_render_image does.  stats: n_scenes entries (one per worker) or NULL.  seconds: wall
 * time of the whole call.  Blocking; the workers' errors come back as this call's.
 * The sum: by default one kernel on scenes[0]'s GPU that reads the peers' buffers in place
 * (hipDeviceEnablePeerAccess; a peer it cannot map is copied over first); with PTX_REDUCE_RCCL in
 * params->pipeline one RCCL reduce over communicators the library creates once per set of GPUs
 * (librccl.so is looked up at run time; PTX_ERR_UNSUPPORTED without it).  Every scene keeps its
 * buffers from call to call: no allocation per image after the first.
 * Processes that hold one GPU each (torch.distributed, MPI) use ptx_render_accumulate with
 * their own sample range and sum the buffers with one RCCL reduce of their own (ptx_reduce_accum).
 */
PtxStatus ptx_render_image_multi(PtxScene* const* scenes, int32_t n_scenes, const PtxRenderParams* params, float* color,
                                 uint8_t* rgb8, double* seconds, PtxStats* stats);

/*
 * The one exchange of the multi-GPU render (no counterpart in the reference, which is one process
 * on one CPU): ncclReduce(sum, float) of `n_floats` accumulation-buffer values onto rank `root` of
 * the caller's communicator (`comm` is an ncclComm_t), queued on `stream`; d_recv may be NULL or
 * equal d_send (in place).  The RCCL library is looked up at run time (librccl.so.1):
 * PTX_ERR_UNSUPPORTED when there is none.
 */
PtxStatus ptx_reduce_accum(void* comm, const float* d_send, float* d_recv, uint64_t n_floats, int root, void* stream);

/*
 * scene.root->intersect(ray) for n rays (framework/renderer.cpp:109, shape.hpp:28).
 * Host arrays; origins/dirs are n*3 floats.  prim = index into PtxSceneDesc::prims or -1;
 * t = HitPoint::distance; pos / normal / raydir = HitPoint::position / surface_normal /
 * ray_direction (framework/hitPoint.hpp:8-16).  Output pointers may be NULL.
 */
PtxStatus ptx_intersect(PtxScene* scene, int64_t n, const float* origins, const float* dirs, int32_t* prim, float* t,
                        float* pos, float* normal, float* raydir);

/*
 * The primary ray of sample (aax, aay) = (sample / aa_steps, sample % aa_steps) of every
 * pixel (framework/renderer.cpp:27-37, :76-82) intersected with the scene: prim (w*h
 * int32) and t (w*h floats), pixel-major y up.  Host arrays, either may be NULL.
 */
PtxStatus ptx_primary_hits(PtxScene* scene, const PtxRenderParams* params, uint32_t sample, int32_t* prim, float* t);

/* PpmWriter::save (framework/ppmwriter.cpp:55-76): ASCII P3 from the flipped rgb8 rows. */
PtxStatus ptx_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/*
 * The same rows as a binary PPM ("P6", 3 bytes per pixel instead of up to 12 characters), and
 * the float colour buffer (y up, as ptx_render_image returns it) as a little-endian PFM
 * ("PF", rows bottom to top, which is the buffer's own order).  No counterpart in the
 * reference, whose only writer is the ASCII one above: at 3840x2160 that text is about
 * 100 MB and takes longer to write than the image takes to render.
 */
PtxStatus ptx_write_ppm_binary(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);
PtxStatus ptx_write_pfm(const char* path, const float* color, uint32_t width, uint32_t height);

/*
 * An image file written BESIDE the rendering (PpmWriter::save, framework/ppmwriter.cpp:55-76,
 * blocks the reference's render thread; at 3840x2160 its text is 90 MB).  d_pixels is a DEVICE
 * buffer: the flipped 8-bit rows of ptx_resolve for the two PPM formats, its float colour buffer
 * for PFM.  The call queues a device-to-host copy into pinned memory on `stream` and returns; a
 * worker thread waits for the copy, formats and writes the file.  The caller may go on using the
 * stream at once (work queued behind the copy may overwrite d_pixels) - e.g. render the next
 * sample range while the preview of the last one is written.  ptx_write_wait joins the worker,
 * frees the job and returns the status of the write; every job must be waited for once.
 */
typedef enum PtxImageFormat { PTX_IMAGE_PPM_ASCII = 0, PTX_IMAGE_PPM_BINARY = 1, PTX_IMAGE_PFM = 2 } PtxImageFormat;
typedef struct PtxWriteJob PtxWriteJob;
PtxStatus ptx_write_image_async(const char* path, PtxImageFormat format, int device, const void* d_pixels, uint32_t width,
                                uint32_t height, void* stream, PtxWriteJob** job);
PtxStatus ptx_write_wait(PtxWriteJob* job);

/*
 * Checkpoint of a render in progress (the reference saves nothing before the image is complete,
 * framework/renderer.cpp:58): the accumulation buffer of ptx_render_accumulate (device, w*h*3
 * floats) together with the range of per-pixel samples it holds, as one little-endian file
 * (32-byte header + the floats), written under a temporary name and renamed.  ptx_accum_load
 * checks that the file is for this image size and aa_steps, fills d_accum and reports the range;
 * rendering the remaining samples into the loaded buffer gives the image an uninterrupted render
 * gives.  Both calls wait for `stream`.
 */
PtxStatus ptx_accum_save(const char* path, int device, const float* d_accum, uint32_t width, uint32_t height, uint32_t aa_steps,
                         uint32_t sample_begin, uint32_t sample_end, void* stream);
PtxStatus ptx_accum_load(const char* path, int device, float* d_accum, uint32_t width, uint32_t height, uint32_t aa_steps,
                         uint32_t* sample_begin, uint32_t* sample_end, void* stream);

/*
 * Profiling level 1: ptx_render_accumulate brackets every bounce-kernel launch with a pair of
 * HIP events on the caller's stream and reports per-depth kernel time in PtxStats (which
 * then must be requested).  Level 2: in addition the traversal kernels run as their counting
 * twins (same code plus one counter per kind of test; slower by the counting) and PtxStats
 * carries trace_tests / shadow_tests.  0 = off, the default.  No counterpart in the
 * reference, whose only timing is a wall clock around render() (framework/renderer.cpp:43,54-56).
 */
PtxStatus ptx_scene_set_profiling(PtxScene* scene, int enable);

/* Introspection for tests and measurements: frames, prims, BVH nodes, BVH depth, device bytes. */
PtxStatus ptx_scene_info(const PtxScene* scene, int32_t* n_frames, int32_t* n_prims, int32_t* n_bvh_nodes,
                         int32_t* max_bvh_depth, uint64_t* device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
