/* rthip.h — C ABI of librthip, the MI355X (gfx950) replacement for the CPU trace loop of
 * Peanutt42/raytracer.
 *
 * The reference has no FFI or plugin interface; its seam is the Rust public API re-exported
 * by src/lib.rs:1-19, and every caller owns the frame loop
 * (raytracer_headless/src/main.rs:34-57, raytracer_runtime/src/main.rs:69-94,
 * raytracer_terminal/src/main.rs:136-150).  This library replaces that loop, not the
 * per-sample call: the scene and camera are handed over once, and one call renders
 * `spp_count` samples of every pixel into an accumulation buffer that lives in HBM.
 *
 * Conventions: every function returns an int32 status (RTHIP_OK = 0), never unwinds or
 * aborts across the ABI, takes plain pointers and sizes, and copies what it needs: the
 * caller owns every host buffer it passes.  A context is bound to one GPU and must be used
 * from one thread at a time.  Calls are synchronous unless noted.
 */
#ifndef RTHIP_H
#define RTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	RTHIP_OK = 0,
	RTHIP_ERR_INVALID = 1,  /* bad argument */
	RTHIP_ERR_EMPTY = 2,    /* empty scene: BVH::new returns None (src/bvh.rs:51) */
	RTHIP_ERR_STATE = 3,    /* scene or camera not set yet */
	RTHIP_ERR_HIP = 4,      /* a HIP runtime call failed; see rthip_last_error */
	RTHIP_ERR_LIMIT = 5,    /* tree deeper than the traversal stack, image too large, out of host memory, ... */
	RTHIP_ERR_INTERNAL = 6  /* a C++ exception was caught at the ABI; see rthip_last_error */
};

enum { RTHIP_SPHERE = 0, RTHIP_CUBE = 1 };                           /* src/scene.rs:36-40 Object */
enum { RTHIP_LAMBERTIAN = 0, RTHIP_METAL = 1, RTHIP_DIELECTRIC = 2 }; /* src/materials.rs:4-8 */
enum { RTHIP_F32 = 0, RTHIP_F64 = 1,
       RTHIP_F32_RAW = 2 /* rthip_trace_primary only: f32 hit records without the f64 refinement */ };
enum { RTHIP_SKY_REFERENCE = 0, RTHIP_SKY_WALLPAPER = 1 };
enum { RTHIP_MEGAKERNEL = 0, RTHIP_WAVEFRONT = 1 };

/* One scene object, in f64 as the reference stores it (src/shapes/sphere.rs:4-9,
 * src/shapes/cube.rs:4-9, src/materials.rs:4-8).  Objects are passed in insertion order;
 * the index in that order is the object id (Scene::add_sphere / add_cube order,
 * src/scene.rs:90-98; spheres then cubes for combine_spheres_and_cubes, src/scenes.rs:5-14).
 * Negative radii and half extents are legal and keep the reference's meaning. */
typedef struct rthip_object {
	int32_t kind;      /* RTHIP_SPHERE | RTHIP_CUBE */
	int32_t material;  /* RTHIP_LAMBERTIAN | RTHIP_METAL | RTHIP_DIELECTRIC */
	double center[3];
	double extent[3];  /* sphere: extent[0] = radius ; cube: half extents */
	double albedo[3];  /* Lambertian, Metal */
	double param;      /* Lambertian: emission ; Metal: fuzz ; Dielectric: ir */
} rthip_object;

/* Arguments of Camera::new (src/camera.rs:25-33).  The library derives the private fields
 * of src/camera.rs:5-11 itself. */
typedef struct rthip_camera_desc {
	double origin[3];
	double direction[3];
	double fov;           /* degrees */
	double focus_dist;
	double defocus_angle; /* degrees; <= 0 disables the defocus offset (src/camera.rs:86) */
	int32_t width, height;
} rthip_camera_desc;

/* One pass of the frame loop (raytracer_headless/src/main.rs:41-55): samples
 * [spp_begin, spp_begin + spp_count) of every pixel this call owns. */
typedef struct rthip_render_params {
	uint32_t spp_begin, spp_count;
	int32_t max_depth;  /* renderer.rs:34 */
	uint32_t seed;      /* key of the counter-based random stream */
	int32_t sky_mode;   /* RTHIP_SKY_* */
	int32_t precision;  /* RTHIP_F32 (production) | RTHIP_F64 (reference arithmetic) */
	int32_t variant;    /* RTHIP_MEGAKERNEL | RTHIP_WAVEFRONT */
	/* Tile partition for multi-GPU runs: this call renders the 8x8 tiles whose row-major
	 * index t satisfies t % part_count == part_index.  1 GPU: part_index 0, part_count 1. */
	int32_t part_index, part_count;
	int32_t reserved[3]; /* must be zero */
} rthip_render_params;

typedef struct rthip_stats {
	uint64_t paths;         /* (pixel, sample) pairs rendered by the last render call */
	uint64_t rays;          /* ray segments (one BVH::trace each) traced by it */
	uint64_t box_tests;     /* filled by instrumented launches only, else 0 */
	uint64_t sphere_tests;
	uint64_t cube_tests;
	double kernel_ms;       /* device time of the last render call's kernels (HIP events) */
	int32_t n_nodes, n_leaves, n_prims, tree_depth; /* of the uploaded scene */
	int32_t scene_in_lds;   /* 1 when the last render kept the scene in LDS */
	int32_t n_gates;        /* gate boxes of the uploaded scene (inverted spheres parted from their outer shell) */
	/* shape of the last render call's megakernel launch (0 for the wavefront variant) */
	int32_t launch_threads;    /* threads per workgroup */
	int32_t launch_grid;       /* workgroups (all resident: the grid is persistent) */
	int32_t launch_lds_bytes;  /* dynamic LDS per workgroup */
	int32_t waves_per_cu;      /* resident waves per CU the launch was planned for */
} rthip_stats;

typedef struct rthip_ctx rthip_ctx;

/* Context on GPU `device` (ordinal as HIP sees it). */
int32_t rthip_create(int32_t device, rthip_ctx** out);
void rthip_destroy(rthip_ctx* ctx);
/* Message for the last non-OK status of this context (or of rthip_create when ctx is NULL). */
const char* rthip_last_error(const rthip_ctx* ctx);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream, which is
 * NOT ordered against the default stream: to launch on the default stream itself (handle 0 in
 * torch) pass hipStreamLegacy.  Waits for what is queued on the stream it leaves. */
int32_t rthip_set_stream(rthip_ctx* ctx, void* hip_stream);

/* Tuning and build knobs (no counterpart in the reference, which has no configuration at all).
 * A context reads the RTHIP_* environment variables ONCE, in rthip_create; this call changes a
 * knob afterwards.  Names are the variable names in lower case without the prefix: launch knobs
 * "scene_lds", "qnodes", "mats_lds" (materials in LDS beside the scene), "acc_lds" (per-wave accumulation
 * tiles in LDS: -1 = scenes with nothing below the root, 0 = never, 1 = wherever they fit), "root_leaf" (0 = a scene that is
 * one leaf with no tree is still walked from its root; 1, the default, tests the leaf at ray start and skips the walk), "packet"
 * (scenes in global memory: 1, the default, walks the 64 camera rays of a batch through the tree together as one wave packet
 * whose node reads are scalar loads; 0 = each ray on its own; same image either way), "overlap" (1, the default: short launches queued with
 * rthip_render_accumulate_async on the context's OWN stream take turns on three streams, so that the tail of one runs under the start of
 * the next; samples are integer adds, so the sum is the same, and every other call on the context is ordered behind all of them;
 * 0 = one launch after the other; with a caller-owned stream launches always follow one another on that stream), "pixel_major" (what such a packet is made of:
 * 1 = all samples of a few neighbouring pixels, 0 = one sample of an 8x8 tile, -1, the default = the former from 65 536 primitives on), "stack_mem"
 * (1 = traversal stacks in global memory instead of LDS; measured slower, off), "xchg" (1 = rays waiting to
 * be shaded are regrouped by kind through queues in LDS, with "xchg_cap", "xchg_pop_min", "xchg_push_max";
 * measured slower, off), "threads", "blocks_per_cu", "grid_reserve", "chunk", "trav_threshold", "inner_exit",
 * "regen_min", "block_order", "wf_batch", "count" take effect with the next render call; build
 * knobs "tree_ref" (1 = device tree in the reference's shape), "tree_gpu" (1 = device tree built
 * on the GPU, a linear BVH with "gpu_leaf" primitives per leaf), "ref_gpu" (tie ranks, gates and primitive
 * records made on the GPU for such uploads and for refits: 1 always, 0 never, -1 = from 2048 objects on), "sah_leaf", "sah_kind_cap" (at most so many
 * spheres and so many cubes per leaf, 0 = no cap), "sah_cprim",
 * "sah_sweep", "big_leaf_min", "no_big_leaf", "no_qnodes", "build_threads", "tie_reverse"
 * (1 = equal sort keys of BVH::new in reversed order, see rthip_bvh_leaf_order) with the next
 * rthip_scene_upload.  Unknown names, and values that are not numbers in the int range, return
 * RTHIP_ERR_INVALID. */
int32_t rthip_set_option(rthip_ctx* ctx, const char* name, double value);

/* Replaces BVH::new(scene) (src/bvh.rs:17-81): builds the reference-shaped median-split
 * tree on the host, flattens it and uploads nodes, primitives and materials to HBM.
 * n == 0 -> RTHIP_ERR_EMPTY. */
int32_t rthip_scene_upload(rthip_ctx* ctx, const rthip_object* objects, int64_t n);
/* No counterpart in the reference, whose scenes never change once BVH::new has run (src/bvh.rs:17-81
 * is its only builder); for callers that move objects between frames.  `objects` are the objects of
 * the last rthip_scene_upload in the same order, at new places, sizes or materials: the same count,
 * each still a sphere or still a cube, the same ones impossible to hit (cubes with a negative half
 * extent), else RTHIP_ERR_INVALID and the scene on the device stays as it was (any OTHER failure of a
 * refit, a HIP error or a limit, leaves the context without a usable scene: upload again).  The device tree keeps its shape
 * and its boxes are fitted again on the GPU, bottom up, one launch per level; tie ranks and gates are those
 * of the reference's tree over the objects as they now are, so every result equals that of a fresh
 * upload.  Far-reaching moves leave a tree that is sound but slow to walk: upload then.
 * `objects` may point into device memory (of the context's GPU) for both calls; a refit of a scene
 * of 2 048 objects and more then runs without a copy to or from the host.  Both calls read such
 * objects on the context's stream only: whatever wrote them must be complete, or queued on that
 * same stream (rthip_set_stream to the producer's stream, or a host wait), before the call. */
int32_t rthip_scene_refit(rthip_ctx* ctx, const rthip_object* objects, int64_t n);
/* Replaces Camera::new (src/camera.rs:25-69).  Changing the image size resets the
 * accumulation buffer; changing only the pose does not (callers reset, as
 * raytracer_runtime/src/main.rs:101-103,149 does). */
int32_t rthip_camera_set(rthip_ctx* ctx, const rthip_camera_desc* cam);
/* Host-only (no GPU needed): the derived fields Camera::new computes — origin, pixel00_loc,
 * pixel_delta_x, pixel_delta_y, defocus_disk_x, defocus_disk_y (6 x 3 values) and
 * defocus_angle (src/camera.rs:5-11,35-68). */
int32_t rthip_camera_fields(const rthip_camera_desc* cam, double out[19]);

/* Replaces the sample loop of raytracer_headless/src/main.rs:41-54 and the per-frame pass of
 * raytracer_runtime/src/main.rs:69-86: adds sum_s sqrt(render(x, y, s)) to the accumulation
 * buffer.  Results do not depend on how samples or tiles are split over calls or GPUs. */
int32_t rthip_render_accumulate(rthip_ctx* ctx, const rthip_render_params* p);
/* Asynchronous form: enqueues on the context's stream and returns.  Statistics are valid
 * after rthip_synchronize.  (With variant RTHIP_WAVEFRONT and max_depth beyond 64 the call waits
 * for the device once every 64 bounces, to see whether any path is left before it queues the next
 * launches; up to 64 bounces it returns without a wait, like the megakernel.) */
int32_t rthip_render_accumulate_async(rthip_ctx* ctx, const rthip_render_params* p);
int32_t rthip_synchronize(rthip_ctx* ctx);
/* The asynchronous render call for callers that cannot allocate device memory (code under HIP
 * graph capture: on a capturing stream the call records only the clearing of its control block and the
 * kernel, no events, and leaves the statistics of rthip_get_stats as they were; tight serving loops;
 * harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  It calls no hipMalloc, hipFree or other allocating routine: the control
 * block of the launch (work-queue counters and statistics, 1 KiB) lives in a 4 KiB array in
 * static device memory that every GPU holds from library load to process exit, instead of the
 * block rthip_create allocates.  Limits: RTHIP_MEGAKERNEL only (the wavefront variant needs
 * gigabytes of path state: RTHIP_ERR_LIMIT); the accumulation buffer of the current image size
 * must already exist (rthip_camera_set or rthip_set_accum_buffer: RTHIP_ERR_STATE otherwise,
 * nothing is allocated for it).  All contexts on one GPU share that GPU's array and the array
 * keeps the last call's contents (the call clears what must start at zero, in stream order):
 * two launches that overlap in time on one GPU would race, so a call made while the last launch
 * through this entry point is still running on ANOTHER stream returns RTHIP_ERR_STATE and launches
 * nothing (calls on one stream follow one another and are fine; replays of a captured graph are not
 * seen by this check and are the caller's to keep apart).  A reduce in flight on the accumulation
 * buffer (rthip_merge_reduce) is waited for on the stream first.  Same results, bit for bit, as
 * rthip_render_accumulate_async. */
int32_t rthip_render_accumulate_noalloc(rthip_ctx* ctx, const rthip_render_params* p);
/* raytracer_runtime/src/main.rs:102-103 (accum_image cleared on a camera change). */
int32_t rthip_reset_accumulation(rthip_ctx* ctx);

/* The accumulation buffer: uint64 words counting units of 2^-36, laid out
 * [tile][channel r,g,b][64 pixels] over the 8x8 tiles of the image in row-major tile order
 * (pixels row-major inside a tile): word = (tile * 3 + channel) * 64 + (y % 8) * 8 + (x % 8).
 * Integer sums make merges across GPUs exact and order-independent.  The pointer is device
 * memory owned by the context; n_words = tiles_x * tiles_y * 192. */
int32_t rthip_accum_buffer(rthip_ctx* ctx, void** device_ptr, uint64_t* n_words);
/* Use caller-owned device memory of n_words uint64 (e.g. a torch tensor, so that it can be
 * handed to an RCCL collective) as the accumulation buffer; NULL returns to the own buffer. */
int32_t rthip_set_accum_buffer(rthip_ctx* ctx, void* device_ptr, uint64_t n_words);
/* Checkpoint / resume of a long render: copy the accumulation words (layout above) to host
 * memory and back.  Sums are integers, so a render resumed from a checkpoint ends bit-identical
 * to an uninterrupted one.  (No counterpart in the reference, whose `output` Vec lives and dies
 * with the process, raytracer_headless/src/main.rs:34-57; its callers can only restart.) */
int32_t rthip_accum_download(rthip_ctx* ctx, uint64_t* words, uint64_t n_words);
int32_t rthip_accum_upload(rthip_ctx* ctx, const uint64_t* words, uint64_t n_words);

/* Multi-GPU merge (no counterpart in the reference, a single-process renderer).  Every GPU
 * renders its share (a tile part or a sample range, see rthip_render_params) into its own
 * full-frame accumulation buffer; the image is the element-wise sum of those buffers.
 *
 * Over RCCL/xGMI, one context per GPU, in one process (one thread per context) or in several:
 * rank 0 makes an id with rthip_comm_unique_id and hands the 128 bytes to the others by any
 * means; every rank calls rthip_comm_init(ctx, id, rank, world) (collective: all ranks must be
 * in the call at once); rthip_merge_reduce then enqueues ONE ncclReduce(sum, uint64) of the
 * active accumulation buffer, in place, behind the renders already queued; root = the rank that
 * ends with the image, or -1 for all of them (all-reduce).  Integer sums: the merged words equal
 * a single GPU's, bit for bit.  The reduce runs on a communication stream of its own: a render
 * queued afterwards into ANOTHER buffer (rthip_set_accum_buffer) overlaps it.  Whatever this
 * library does next to the SAME buffer (render, reset, read-back, another reduce) waits for it
 * first, and so does rthip_synchronize; a caller that touches the buffer itself (a memset on its
 * own stream) calls rthip_merge_wait(ctx, buffer) before, which orders the context's stream
 * behind the reduces in flight on that buffer (NULL: on any buffer).  librccl is looked up at
 * the first call (RTHIP_ERR_STATE when the host has none). */
int32_t rthip_comm_unique_id(uint8_t id[128]);
int32_t rthip_comm_init(rthip_ctx* ctx, const uint8_t id[128], int32_t rank, int32_t world);
int32_t rthip_merge_reduce(rthip_ctx* ctx, int32_t root);
int32_t rthip_merge_wait(rthip_ctx* ctx, const void* device_ptr);
/* The same sum without RCCL, for contexts of ONE process called from one thread (several GPUs,
 * or several contexts on one GPU): adds the accumulation buffers of ctxs[0..n) into
 * ctxs[root]'s (peer copies over xGMI where the GPUs differ) and waits for the result. */
int32_t rthip_merge_local(rthip_ctx* const* ctxs, int32_t n, int32_t root);

/* Read back sum_s sqrt(sample) per channel, row-major [height][width][3] (the Vec<Vec3>
 * `output` of raytracer_headless/src/main.rs:34 before the division of :55). */
int32_t rthip_read_accum(rthip_ctx* ctx, double* rgb);
/* raytracer_headless/src/main.rs:6-12,55: (accum / spp_total * 255.0) as u8, row-major RGB8. */
int32_t rthip_read_rgb8(rthip_ctx* ctx, uint32_t spp_total, uint8_t* rgb);
/* raytracer_runtime/src/main.rs:7-17,154-160: 0x00RRGGBB per pixel. */
int32_t rthip_read_xrgb32(rthip_ctx* ctx, uint32_t spp_total, uint32_t* xrgb);

/* Parity probe for BVH::trace (src/bvh.rs:130-142): one ray per pixel.  jitter = 0: from the
 * camera origin through the pixel centre; jitter = 1: the ray Camera::get_ray produces for
 * sample `sample` (src/camera.rs:83-92).  Outputs are host arrays, row-major; id = -1, t = 0
 * on a miss.  t and normal are written as doubles in both precisions. */
int32_t rthip_trace_primary(rthip_ctx* ctx, int32_t precision, int32_t jitter, uint32_t seed, uint32_t sample,
                            int32_t* id, double* t, double* normal, uint8_t* front_face);
/* Debug probe: linear radiance of every sample of render() (src/renderer.rs:29-40), host
 * array [spp_count][height][width][3] of doubles, nothing accumulated. */
int32_t rthip_render_samples(rthip_ctx* ctx, const rthip_render_params* p, double* samples);

int32_t rthip_get_stats(rthip_ctx* ctx, rthip_stats* out);
/* Host-only (no GPU needed): builds the tree of BVH::new (src/bvh.rs:17-81) and reports its
 * shape: object ids leaf by leaf, left to right (reference order inside a leaf), the size of
 * every leaf, the leaf count and the depth in inner-node levels.  ids holds n entries,
 * leaf_sizes at most n; any output pointer may be NULL. */
int32_t rthip_bvh_leaf_order(const rthip_object* objects, int64_t n, int32_t* ids, int32_t* leaf_sizes, int64_t* n_leaves,
                             int32_t* depth);

/* Host-only (no GPU needed): how many gate boxes the build attaches, and to how many spheres.
 * A sphere of negative radius caches an inverted box (src/shapes/sphere.rs:13-19) that adds
 * nothing to the boxes BVH::new unions above it (src/bvh.rs:53-56,68); where a split parts it
 * from its outer shell, the reference reaches it only through boxes smaller than the sphere,
 * and those boxes travel with it as gates.  Any output pointer may be NULL. */
int32_t rthip_bvh_gate_count(const rthip_object* objects, int64_t n, int64_t* n_gates, int64_t* n_gated_spheres);

/* Host-only self-check (no counterpart in the reference: its tree is never wrong by construction,
 * this one is conservative by rounding rules that deserve a test).  Builds the device tree as
 * rthip_scene_upload would and verifies that every f32 child box, and every quantised one,
 * contains the f32 bounds of all primitives below it, that every object that can be hit sits
 * in exactly one leaf, and that the depth is as reported.  violations = 0 and RTHIP_OK when
 * sound; otherwise RTHIP_ERR_STATE and rthip_last_error(NULL) describes the first violation.
 * checksum receives a hash of every array the device would be given (the build is
 * deterministic, on one thread or several).  Any output pointer may be NULL. */
int32_t rthip_bvh_validate(const rthip_object* objects, int64_t n, int64_t* violations, int32_t* n_nodes, int32_t* depth,
                           uint64_t* checksum);

/* (No counterpart in the reference either.)  The same checks on the tree a context holds on the
 * device, downloaded for the purpose: the one
 * rthip_scene_upload built on the host, or, with the build option "tree_gpu", the linear BVH the
 * GPU built (Morton codes, radix sort, hierarchy and boxes in kernels; the tie ranks and gates that
 * hang on the reference's tree come from level-by-level segmented sorts on the GPU as well, see the
 * option "ref_gpu").  Any output pointer may be NULL. */
int32_t rthip_scene_validate(rthip_ctx* ctx, int64_t* violations, int32_t* n_nodes, int32_t* depth);

/* Seeded scene factories (no GPU needed).  Same distributions as the reference's factories,
 * which draw from an OS-seeded generator and are therefore not reproducible themselves:
 *   "default"   Scene::create_sample_scene          src/scene.rs:122-205
 *   "simple"    create_simple_scene                 src/scenes.rs:17-75
 *   "glass"     create_glass_scene                  src/scenes.rs:78-121
 *   "metal"     create_10_metallic_scene            src/scenes.rs:124-170
 *   "wallpaper" create_wallpaper_scene              src/scenes.rs:173-198
 *   "sample"    scenes::create_sample_scene         src/scenes.rs:201-300 (many spheres and cubes)
 *   "synthetic:N"  N random spheres and cubes (no counterpart in the reference)
 * Writes at most `capacity` objects and returns the scene's object count in *n (call with
 * capacity 0 to size the buffer). */
int32_t rthip_scene_generate(const char* name, uint32_t seed, rthip_object* out, int64_t capacity, int64_t* n);
/* get_camera_rotation (src/camera.rs:95-103). */
void rthip_camera_rotation(double yaw_deg, double pitch_deg, double out[3]);

#ifdef __cplusplus
}
#endif
#endif /* RTHIP_H */
