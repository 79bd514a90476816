/*
 * rthip.h -- C ABI of the MI355X (gfx950) rendering backend.
 *
 * The library behind this header replaces ONE path of patricklbell/raytracer:
 * LBVH build -> TLAS/BLAS traversal -> ray/triangle intersection -> shading ->
 * framebuffer.  Every entry point below names the reference interface it stands
 * in for (paths relative to the reference's repository root).
 *
 * Conventions
 *   - plain C, POD arguments only; no HIP or torch types in any signature
 *     (streams travel as `void*` holding a hipStream_t, NULL = default stream);
 *   - every function returns 0 on success and a non-zero RTHIP_E_* code on
 *     failure; rthip_last_error() then returns a human readable message;
 *   - "host" pointers are ordinary CPU memory, "device" pointers (`d_` prefix)
 *     are HBM addresses valid on the context's GPU;
 *   - arrays of meshes / instances / materials are passed in the order in which
 *     the reference iterates its world lists (`first -> next`), i.e. element 0
 *     is the item that was added LAST (src/raytracer/raytracer_core.c:21,37,53
 *     push to the front).  TLAS ids are 1-based indices into that order,
 *     exactly like `id = idx + 1` in src/lbvh/lbvh.c:116.
 *   - one context drives one GPU.  Multi-GPU jobs either run one process (one context)
 *     per GPU and shard with rthip_shard, or make a group of contexts in one process
 *     (rthip_group_*), which shards and gathers by itself.
 *   - a context is not thread-safe: its calls must come from one thread at a time
 *     (the reference's callers are single-threaded, demos_inc.c:9-10).  Different
 *     contexts are independent.  The *_device entry points only enqueue work on
 *     the given stream; builds (rthip_blas_commit, rthip_tlas_build) wait for the
 *     GPU, so a cast enqueued after them on any stream sees the finished trees.
 *     The builder's workspace belongs to the context: two builds on different
 *     streams of one context must not overlap.  Casts and ray queries may: frames
 *     enqueued on different streams run concurrently (one frame's first waves fill
 *     the GPU while another's last waves drain; measured 1.1x to 2.8x the frames
 *     per second of one stream), each into its own framebuffer, and not with
 *     RTHIP_KERNEL_WAVEFRONT or with counting on, whose buffers belong to the
 *     context.  At most 64 launches of the persistent kernels (RTHIP_KERNEL_PERSISTENT,
 *     long ray streams) may be unfinished at a time: the 65th returns RTHIP_E_LIMIT
 *     (each owns a slot of work counters, reused only after an event recorded behind
 *     its launch has completed).  A cast on a stream that is being captured becomes
 *     graph nodes (kernels only; RTHIP_KERNEL_AUTO skips its probe then); the graph
 *     may be replayed as long as the scene is not rebuilt, beside eager casts (captured
 *     launches of the persistent kernels get work-counter slots of their own, at most
 *     192 per scene build), but two replays of one graph must not overlap each other.
 */
#ifndef RTHIP_H
#define RTHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTHIP_ABI_VERSION 2

enum {
    RTHIP_OK = 0,
    RTHIP_E_INVALID = 1,  /* bad argument / call order                      */
    RTHIP_E_HIP = 2,      /* a HIP runtime call failed                      */
    RTHIP_E_NOMEM = 3,    /* device or host allocation failed               */
    RTHIP_E_LIMIT = 4     /* input exceeds a documented limit               */
};

/* RT_MaterialType, src/raytracer/raytracer_core.h:65-72 (same numeric values) */
enum {
    RTHIP_MAT_LAMBERTIAN = 0,
    RTHIP_MAT_DIELECTRIC = 1,
    RTHIP_MAT_METAL = 2,
    RTHIP_MAT_NORMAL = 3,
    RTHIP_MAT_LIGHT = 4
};

/* RT_InstanceType, src/raytracer/raytracer_core.h:31-35 (same numeric values) */
enum { RTHIP_INST_SPHERE = 0, RTHIP_INST_MESH = 1 };

/* RT_Material, src/raytracer/raytracer_core.h:74-84 */
typedef struct rthip_material {
    uint32_t type;
    float albedo[3];
    float emissive[3];
    float roughness;
    float ior;
    uint32_t billboard;
} rthip_material;

/* RT_Instance (+ RT_SphereInstance / RT_MeshInstance),
 * src/raytracer/raytracer_core.h:17-45.  `material` is the index of the
 * material in the array given to rthip_tlas_build, -1 for the reference's zero
 * handle (src/raytracer/cpu/raytracer_cpu.c:264-266).  `blas_id` is the value
 * the reference stores in RT_Mesh.blas_id (raytracer_cpu.c:112). */
typedef struct rthip_instance {
    uint32_t type;
    int32_t material;
    uint32_t blas_id;
    float sphere_center[3];
    float sphere_radius;
    float translation[3];
    float rotation[4]; /* quaternion x,y,z,w */
    float scale[3];
} rthip_instance;

/* RT_CastSettings, src/raytracer/raytracer_core.h:163-178 */
typedef struct rthip_cast_settings {
    float eye[3];
    float up[3];
    float forward[3];
    float right[3];
    float viewport[3];
    uint32_t samples; /* per axis: samples*samples paths per pixel */
    float ior;
    uint32_t defocus;
    float defocus_disk[2];
    uint32_t orthographic;
} rthip_cast_settings;

/* One LBVH node in depth-first pre-order.  The left child of an inner node is
 * the next record; `skip` is the index of the first record after the node's
 * subtree (== node count for the last subtree).  `id` is 0 for inner nodes and
 * the 1-based primitive id for leaves: the same value as LBVH_Node.id,
 * src/lbvh/lbvh.h:4-9.  Serialising records in array order reproduces the
 * reference's dump (src/lbvh/extra/dump.c:1-13) up to its null-child bytes. */
typedef struct rthip_node {
    float aabb_min[3];
    uint32_t skip;
    float aabb_max[3];
    uint32_t id;
} rthip_node;

/* How one context's share of a frame is chosen when several GPUs render it.
 * mode 0: everything.  mode 1 (tiles): pixels are cut into 64x64 tiles numbered
 * row-major; this context renders tiles with (tile % world) == rank and leaves
 * other pixels at 0, so the shares sum to the frame.  mode 2 (samples): this
 * context renders sample indices s with (s % world) == rank of every pixel and
 * writes the plain sum of their radiance (no 1/spp factor).  mode 3 (bands): the
 * frame is cut into `world` bands of whole rows, ceil(height / world) rounded up to
 * a multiple of 64 rows each (the last bands may be short or empty); this context
 * renders band `rank` and does NOT touch the other rows, so several launches can
 * fill one framebuffer (rthip_cast does, to send finished rows to the host while the
 * rest still renders) and a band is one contiguous block of memory to ship. */
typedef struct rthip_shard {
    uint32_t mode;
    uint32_t rank;
    uint32_t world;
} rthip_shard;

/* Counters of the most recent cast / trace call and timings of the most recent
 * build, read back on demand. */
typedef struct rthip_stats {
    uint64_t rays;          /* TLAS closest-hit queries (primary + bounces)   */
    uint64_t nodes_tested;  /* BVH nodes whose box was tested (both levels)   */
    uint64_t tris_tested;   /* triangles handed to the triangle test          */
    uint64_t inst_entered;  /* TLAS leaves entered                            */
    uint64_t wave_steps;    /* traversal-loop iterations summed over waves    */
    float last_cast_ms;     /* device time of the last cast / trace kernel(s) */
    float last_build_ms;    /* device time of the last LBVH build             */
    uint32_t last_cast_kernel; /* RTHIP_KERNEL_* the last cast ran (never AUTO)  */
    uint32_t last_cast_refill; /* persistent kernel: idle lanes that end a traversal phase; else 0 */
    float last_encode_ms;   /* device time of the last rthip_hdr_encode       */
    uint32_t reserved;
    uint64_t lane_steps;    /* the same iterations summed over the lanes that took part in them:
                               lane_steps / (64 * wave_steps) = share of the lanes at work, in either
                               traversal mode (a step of the ordered walk tests up to four boxes, so
                               nodes_tested is no measure of it there) */
} rthip_stats;

typedef struct rthip_ctx rthip_ctx;

/* ---- life cycle ------------------------------------------------------- */

/* rt_make_tracer, src/raytracer/cpu/raytracer_cpu.c:14-25 (allocation part).
 * Fails (RTHIP_E_HIP) when no gfx950 device `device_id` is usable: there is no
 * CPU fallback. */
int rthip_create(rthip_ctx** out_ctx, int device_id);
/* rt_tracer_cleanup, raytracer_cpu.c:38-43 */
void rthip_destroy(rthip_ctx* ctx);
/* No counterpart (the reference hooks return void and assert). */
const char* rthip_last_error(const rthip_ctx* ctx);
int rthip_abi_version(void);

/* RT_TracerSettings as copied by rt_make_tracer, raytracer_cpu.c:20-21
 * (winding order is never copied there, so it does not exist here). */
int rthip_set_tracer(rthip_ctx* ctx, uint32_t max_bounces, int sky);

/* Which bounds normalise the centroids before they are quantised to Morton codes, for every tree built afterwards
 * (BLAS and TLAS): the reference's release builds take the bounds of the centroids (lbvh.c:97-100), its BUILD_DEBUG
 * builds the bounds of the boxes (lbvh.c:101-104), so the two flavours build different trees from the same input.
 * Default RTHIP_BOUNDS_CENTROIDS; a backend compiled into a BUILD_DEBUG build selects RTHIP_BOUNDS_BOXES to write the
 * out.bvh that build writes. */
enum { RTHIP_BOUNDS_CENTROIDS = 0, RTHIP_BOUNDS_BOXES = 1 };
int rthip_set_morton_bounds(rthip_ctx* ctx, int mode);

/* ---- acceleration structures ------------------------------------------ */

/* rt_tracer_build_blas -> rt_cpu_build_blas, raytracer_cpu.c:26-31,101-116.
 * begin discards any previous BLAS set (arena_clear, :29); add_mesh copies one
 * mesh to HBM (the caller's memory is not retained); commit builds one LBVH
 * per mesh (rt_cpu_blas_node_from_mesh, :57-99 + lbvh_make, src/lbvh/lbvh.c:92).
 * `verts` is the reference's interleaved vertex buffer (16 bytes per present
 * attribute, src/geo/geo.c:2-15); nrm_off < 0 means "no normal attribute";
 * idx == NULL or index_count == 0 means auto-index (raytracer_cpu.c:59). */
int rthip_blas_begin(rthip_ctx* ctx, uint32_t n_meshes);
int rthip_blas_add_mesh(rthip_ctx* ctx, uint32_t blas_id, const void* verts, uint32_t vertex_count,
                        uint32_t stride_bytes, uint32_t pos_off, int32_t nrm_off, const uint32_t* idx,
                        uint32_t index_count);
int rthip_blas_commit(rthip_ctx* ctx);

/* rt_tracer_build_tlas -> rt_cpu_build_tlas, raytracer_cpu.c:32-37,149-174:
 * world AABB per instance (:131-147, :583-606) then lbvh_make over them. */
int rthip_tlas_build(rthip_ctx* ctx, const rthip_instance* instances, uint32_t n_instances,
                     const rthip_material* materials, uint32_t n_materials);

/* ---- rendering --------------------------------------------------------- */

/* rt_tracer_cast -> rt_cpu_raygen, raytracer_cpu.c:45-49,179-242.  out_rgb is
 * host memory, width*height*3 floats, row-major, y down, fully overwritten.
 * `seed` replaces the reference's global srand() state (src/demos/demos_inc.c:97):
 * random numbers come from a counter-based generator keyed by (seed, pixel,
 * sample, draw index), so the image does not depend on scheduling or GPU count. */
int rthip_cast(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* out_rgb, int width, int height);

/* Optional: page-locked host memory for the framebuffer (hipHostMalloc).  rthip_cast renders frames of 2^19 pixels or
 * more in four bands of rows (rthip_shard mode 3; RTHIP_CAST_BANDS=1..8), each on a stream of its own with its copy
 * behind it, so that finished rows travel while the rest renders.  Into a buffer from rthip_host_alloc the bands go by
 * DMA at the link's rate (a 1920x1080 call: 0.72 ms); into ordinary (pageable) memory they go by DMA into page-locked
 * staging of the context, a quarter of a band at a time, and from there by host threads, two per band (0.82 ms; the runtime's own staged copy of the
 * whole frame, which RTHIP_CAST_BANDS=1 keeps, makes it 1.85 ms).  Free with rthip_host_free, or implicitly by
 * rthip_destroy.  No counterpart in the reference (its framebuffer is arena memory and never leaves the CPU,
 * src/demos/cornell/main.c:214, raytracer_cpu.c:190-192); page-locking the caller's own allocation (hipHostRegister)
 * is not offered because not every host allows it. */
int rthip_host_alloc(rthip_ctx* ctx, uint64_t bytes, void** out_ptr);
int rthip_host_free(rthip_ctx* ctx, void* ptr);

/* Same work with the framebuffer in HBM and no host synchronisation: the call
 * only enqueues on `stream`.  With shard->mode != 0 the buffer holds this
 * context's share as described at rthip_shard.  shard == NULL means mode 0. */
int rthip_cast_device(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* d_out_rgb, int width,
                      int height, const rthip_shard* shard, void* stream);
/* Progressive rendering: the same, but the frame is ADDED to what d_accum_rgb holds (pixels of other ranks' tiles are
 * left alone), so that frames of different seeds sum up in one buffer without a pass of their own; the caller divides
 * by the number of frames in the end.  The reference gets more samples only through `samples` of one cast
 * (raytracer_cpu.c:196-236); no counterpart there.  Accumulating casts into ONE buffer must not overlap each other
 * (use one accumulator per stream). */
int rthip_cast_accumulate_device(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* d_accum_rgb,
                                 int width, int height, const rthip_shard* shard, void* stream);

/* ---- several GPUs of one node in one process ------------------------------ */

/* The reference has one tracer and no notion of devices (raytracer_core.h:180-184); a host that wants the whole node
 * behind those five hooks makes ONE group instead of one context.  A group holds one context per entry of device_ids
 * (an id may repeat: contexts then share a GPU, which is how the path is tested on a one-GPU box).  The scene calls
 * mirror the single-context ones and replicate the scene on every device (one host thread per device, started with
 * the group and ended with it; every device builds its own BVHs from the same host data).  rthip_group_cast renders
 * ONE frame with all devices:
 *   shard_mode 1 (tiles)    device k renders the 64x64 tiles t with t % n == k and ships only those (1/n of the frame),
 *                           packed: straight to the host buffer (rthip_group_cast: each device copies its own tiles over
 *                           its own PCIe link into page-locked staging and its thread places the rows), or over xGMI to
 *                           the first device (rthip_group_cast_device).  Bit-identical to rthip_cast.
 *   shard_mode 2 (samples)  device k renders the samples s with s % n == k of every pixel; the partial sums are added
 *                           on the first device, then scaled by 1/samples^2: equal to rthip_cast up to the rounding of
 *                           another order of additions.
 * Transport between the devices (rthip_group_transport names the one in use): RCCL -- one communicator per device
 * (ncclCommInitAll), the tiles by grouped ncclSend / ncclRecv, the partial sums by ncclReduce -- when the group's
 * devices are distinct and the RCCL library can be loaded at run time (it is not linked); otherwise
 * hipMemcpyPeerAsync into the first device and a sum there in rank order.  RTHIP_GROUP_TRANSPORT=peer|rccl in the
 * environment insists on one of them (rccl: group creation fails without it, and a group of one device runs its
 * exchange through RCCL as well).
 * d_out_rgb of rthip_group_cast_device is memory of the first device (rthip_group_ctx(g, 0)); both calls return when
 * the frame is complete.  Settings that are per context (cast kernel, counting, statistics, taps, tree read-back) go
 * through rthip_group_ctx.  A group is driven from one thread at a time, like a context. */
typedef struct rthip_group rthip_group;
int rthip_group_create(rthip_group** out_group, int n, const int* device_ids);
void rthip_group_destroy(rthip_group* group);
int rthip_group_size(const rthip_group* group);
rthip_ctx* rthip_group_ctx(rthip_group* group, int rank);
const char* rthip_group_last_error(const rthip_group* group);
const char* rthip_group_transport(const rthip_group* group);
int rthip_group_set_tracer(rthip_group* group, uint32_t max_bounces, int sky);
int rthip_group_set_morton_bounds(rthip_group* group, int mode);
int rthip_group_set_traversal(rthip_group* group, int mode);
int rthip_group_blas_begin(rthip_group* group, uint32_t n_meshes);
int rthip_group_blas_add_mesh(rthip_group* group, uint32_t blas_id, const void* verts, uint32_t vertex_count,
                              uint32_t stride_bytes, uint32_t pos_off, int32_t nrm_off, const uint32_t* idx,
                              uint32_t index_count);
int rthip_group_blas_commit(rthip_group* group);
int rthip_group_tlas_build(rthip_group* group, const rthip_instance* instances, uint32_t n_instances,
                           const rthip_material* materials, uint32_t n_materials);
int rthip_group_cast(rthip_group* group, const rthip_cast_settings* cast, uint64_t seed, float* out_rgb, int width,
                     int height, int shard_mode);
int rthip_group_cast_device(rthip_group* group, const rthip_cast_settings* cast, uint64_t seed, float* d_out_rgb, int width,
                            int height, int shard_mode);

/* The tile exchange's two kernels, for hosts that move the tiles themselves (one process per GPU, a gather over RCCL):
 * rank r's tiles are the 64x64 tiles t with t % world == r in row-major tile order (rthip_shard mode 1);
 * rthip_tiles_of_rank is how many there are.  Packed, tile s of the rank occupies floats [s * 12288, (s + 1) * 12288):
 * 64 rows of 64 rgb pixels, the part outside the image zero.  pack reads them out of a frame, unpack places them into
 * one (other pixels untouched).  Both enqueue on `stream` and return.  No counterpart in the reference. */
uint32_t rthip_tiles_of_rank(int width, int height, uint32_t rank, uint32_t world);
int rthip_tiles_pack_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint32_t rank, uint32_t world,
                            float* d_packed, void* stream);
int rthip_tiles_unpack_device(rthip_ctx* ctx, const float* d_packed, int width, int height, uint32_t rank, uint32_t world,
                              float* d_rgb, void* stream);

/* ---- parity and measurement taps -------------------------------------- */

/* lbvh_make, src/lbvh/lbvh.c:92-125, on caller-supplied boxes (n x {min[3],max[3]}).
 * Any output pointer may be NULL.  out_codes[i] is the Morton code of box i
 * (lbvh_c_to_morton_code, lbvh.c:74-90, release-mode centroid bounds :97-100);
 * out_codes_sorted / out_ids_sorted are the sorted NodeIndex array (lbvh.c:5-8,120;
 * ids 1-based; equal codes keep ascending id); out_nodes receives 2n-1 pre-order
 * records.  out_ms = device time of the build kernels. */
int rthip_lbvh_build(rthip_ctx* ctx, const float* aabb6, uint64_t n, uint64_t* out_codes,
                     uint64_t* out_codes_sorted, uint32_t* out_ids_sorted, rthip_node* out_nodes,
                     float* out_ms);

/* The same build with the boxes already in HBM and the tree left in HBM: what the
 * build benchmark times.  d_nodes must hold 2n-1 records.  Enqueues on `stream`
 * and returns without synchronising, with one exception: a build larger than any before it first grows the
 * context's workspace (see rthip_lbvh_reserve).  Builds of more than 131 072 boxes sort by the upper half of the Morton
 * codes and finish inside the stretches of keys that share it, all on the device; whether a stretch was long enough to
 * need the slow finish travels to the host behind the build and makes the NEXT build of that many boxes sort by all
 * eight bytes (RTHIP_SORT_TOP_BYTES=8 in the environment does so from the start).  A stream that is being captured
 * records the same kernels.
 * Hand-offs between workgroups: builds of more than 131 072 boxes finish the nodes that span chunks of leaves with a
 * chain of device-wide hand-offs.  Their fast form (records and counters as device-scope atomics, ordered by waiting for
 * the stores) is how gfx950 behaves, not something the HIP memory model promises, so a context uses it only after it has
 * checked it on its own device: before its first such build it builds one tree of 400 000 boxes in both forms (a few
 * milliseconds, once per context, the calling thread waits) and keeps the memory model's release / acquire form -- 8 %
 * slower at 10^7 boxes -- on any difference, and for builds that cannot wait for the check (a first large build on a
 * stream that is being captured).  RTHIP_REFIT_FENCE=0|1|2 in the environment chooses a form outright (0 release /
 * acquire, 1 __threadfence, 2 the fast one). */
int rthip_lbvh_build_device(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, rthip_node* d_nodes, void* stream);

/* Grows the builder's workspace to what builds of up to n boxes (or triangles) need.  rthip_lbvh_build_device and
 * rthip_mesh_build_device grow it themselves when a build is larger than any before it -- a hipFree / hipMalloc inside
 * the call, which waits for the device and is not allowed while a stream is being captured; after rthip_lbvh_reserve(n)
 * builds of at most n boxes only enqueue.  No counterpart in the reference (its arenas grow on demand, lbvh.c:48). */
int rthip_lbvh_reserve(rthip_ctx* ctx, uint64_t n);

/* rthip_lbvh_build_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same tree bit for bit, but
 * the builder's workspace is a fixed array inside the library (about 17 MB per GPU, held until the process ends) instead
 * of memory grown on demand: the call makes no hipMalloc / hipFree, and n is limited to RTHIP_NOALLOC_MAX_BOXES
 * (RTHIP_E_LIMIT beyond).  All contexts on one GPU share that array: calls that overlap on one GPU (other streams,
 * other contexts, other threads) race; calls on one stream, or ordered by events, do not. */
#define RTHIP_NOALLOC_MAX_BOXES (1u << 17)
int rthip_lbvh_build_device_noalloc(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, rthip_node* d_nodes, void* stream);

/* rt_cpu_blas_node_from_mesh, raytracer_cpu.c:57-99, for one mesh whose vertex
 * positions (SoA: d_px, d_py, d_pz) and indices are already in HBM: triangle
 * boxes + lbvh_make.  d_indices == NULL means auto-index.  d_nodes must hold
 * 2*n_tris-1 records. */
int rthip_mesh_build_device(rthip_ctx* ctx, const float* d_px, const float* d_py, const float* d_pz,
                            uint32_t vertex_count, const uint32_t* d_indices, uint64_t n_tris,
                            rthip_node* d_nodes, void* stream);

/* One closest-hit query per pixel with the pixel-centre ray that rt_cpu_raygen
 * generates for samples == 1 (raytracer_cpu.c:195-231, no defocus jitter):
 * lbvh_query_ray on the TLAS with rt_cpu_tlas_hit (raytracer_cpu.c:402-407).
 * out_instance_id = 1-based TLAS leaf id (0 = miss), out_tri_idx = index-buffer
 * offset of the triangle ((id-1)*3, raytracer_cpu.c:467,479; 0 for spheres and
 * misses), out_t = interval.max after the query (-1.0f on a miss). */
int rthip_primary_hits(rthip_ctx* ctx, const rthip_cast_settings* cast, uint32_t* out_instance_id,
                       uint32_t* out_tri_idx, float* out_t, int width, int height);

/* rt_cpu_intersect, raytracer_cpu.c:402-457, for n caller-supplied rays (host
 * arrays of 3 floats each; directions are used as given).  Outputs as for
 * rthip_primary_hits; out_uv (2 floats per ray, may be NULL) receives the
 * barycentrics stored by geo_intersect_tri (src/geo/geo.c:106-107); out_normal
 * (3 floats per ray, may be NULL) the shading normal of raytracer_cpu.c:421-453. */
int rthip_trace_rays(rthip_ctx* ctx, const float* origins, const float* directions, uint64_t n,
                     uint32_t* out_instance_id, uint32_t* out_tri_idx, float* out_t, float* out_uv,
                     float* out_normal);

/* The same query with rays and results in HBM (SoA not required: o and d are n x 3
 * floats); enqueues on `stream` without synchronising.  d_hits is n x 4 uint32:
 * {instance id, tri_idx, t bits, 0}. */
int rthip_trace_rays_device(rthip_ctx* ctx, const float* d_origins, const float* d_directions, uint64_t n,
                            uint32_t* d_hits, void* stream);

/* Occlusion ("shadow") rays: out[i] = 1 when the TLAS query of rt_cpu_intersect
 * (raytracer_cpu.c:402-407: lbvh_query_ray with rt_cpu_tlas_hit) run over the interval
 * [EPSILON_F32, tmax] instead of [EPSILON_F32, MAX_F32] would report a hit, else 0.  The
 * reference itself casts no shadow rays; this is that same query stopped at the first
 * accepted primitive.  Directions need not be normalised (t is in units of |direction|). */
int rthip_occluded(rthip_ctx* ctx, const float* origins, const float* directions, uint64_t n, float tmax, uint8_t* out);
int rthip_occluded_device(rthip_ctx* ctx, const float* d_origins, const float* d_directions, uint64_t n, float tmax,
                          uint8_t* d_out, void* stream);

/* How the BVHs are walked by every cast and ray query that follows.  No counterpart in the reference (one walk,
 * lbvh.c:151-161).
 *   RTHIP_TRAV_EXACT (default)  the reference's walk: depth first, always the left child before the right one,
 *                               whatever the ray.  Results are the reference's bit for bit.
 *   RTHIP_TRAV_FAST             opt-in: the same trees (collapsed two levels at a time into 4-wide records, every box
 *                               a box of the binary tree), the same box / triangle / sphere arithmetic and the same
 *                               acceptance rule (min <= t <= max), but the nearest box is entered first; the others
 *                               wait on a per-lane stack in LDS with their entry distance and are dropped unseen
 *                               once a closer hit has been found.  A ray's chain of dependent fetches is half as
 *                               long and it tests fewer boxes.  The result is the closest hit as well;
 *                               it differs from the reference's only where the reference's answer is an artefact of
 *                               its visiting order: which of several primitives at exactly the same distance is
 *                               reported, and seam rays where the reference skips a triangle because its box (tested
 *                               without slack) lies behind the interval already shrunk to a neighbour's hit while
 *                               the triangle test (with slack) would have accepted it, or the other way round.
 *                               The measured mismatch counts are in the design notes.  The wavefront kernel has no
 *                               fast walk: RTHIP_KERNEL_WAVEFRONT renders with RTHIP_KERNEL_PERSISTENT in this mode.
 * Switching to FAST builds the extra records for the trees that exist (and for every tree built afterwards: room for
 * 128 bytes per inner node, about half of it used); it waits for the GPU. */
enum { RTHIP_TRAV_EXACT = 0, RTHIP_TRAV_FAST = 1 };
int rthip_set_traversal(rthip_ctx* ctx, int mode);

/* Reads (and clears) the traversal counters; blocks until the context's work is done. */
int rthip_get_stats(rthip_ctx* ctx, rthip_stats* out);
/* Turns traversal counting on or off (off by default: counting costs atomics). */
int rthip_set_counting(rthip_ctx* ctx, int enabled);

/* Which kernel rthip_cast / rthip_cast_device run.  All of them produce the same bits.  No
 * counterpart in the reference (its renderer is one loop, raytracer_cpu.c:179-242).
 *   RTHIP_KERNEL_AUTO        default.  Frames of 2^20 paths and more: a probe first renders a sparse sample of
 *                            tiles with traversal counters on (into the same framebuffer) and the kernel
 *                            follows from the measured lane utilisation and walk length; the choice is kept
 *                            for this scene and frame geometry (the first such call waits on the stream;
 *                            skipped under stream capture).  Smaller frames: static for small scenes,
 *                            persistent for deep ones
 *   RTHIP_KERNEL_PERSISTENT  resident waves, lanes take a new path as soon as theirs ends
 *   RTHIP_KERNEL_STATIC      one pixel per thread (a kernel selected by name never probes: the call only enqueues)
 *   RTHIP_KERNEL_STATIC_SAMPLES  one SAMPLE per thread, the samples of a pixel in neighbouring lanes and summed
 *                            in sample order (samples = 2, 4 or 8; any other count runs RTHIP_KERNEL_STATIC).
 *                            AUTO uses it wherever it would run STATIC, and without a probe for frames of
 *                            fewer than 40 000 8x8 pixel tiles (20 480 for scenes of >= 64 instances or
 *                            >= 2^20 triangles)
 *   RTHIP_KERNEL_WAVEFRONT   ray generation, traversal, shading and resolve as separate kernels
 * The environment variable RTHIP_CAST=auto|persistent|static|static-samples|wavefront sets the initial value. */
enum {
    RTHIP_KERNEL_PERSISTENT = 0,
    RTHIP_KERNEL_STATIC = 1,
    RTHIP_KERNEL_WAVEFRONT = 2,
    RTHIP_KERNEL_AUTO = 3,
    RTHIP_KERNEL_STATIC_SAMPLES = 4
};
int rthip_set_cast_kernel(rthip_ctx* ctx, int kernel);
/* Scheduling parameters of RTHIP_KERNEL_PERSISTENT when it is selected by name (AUTO picks its own): `refill`
 * in 1..64 = lanes with no ray to walk that make a wave stop walking and shade / fetch new paths (default 16);
 * `inst_batch` in 0..64 = lanes parked at an instance that make the wave enter the instances, 0 = a lane enters
 * at once (default 0).  They change the schedule only, never a bit of the result.  No counterpart in the reference. */
int rthip_set_persistent_params(rthip_ctx* ctx, uint32_t refill, uint32_t inst_batch);

/* TLAS / BLAS tree read-back for dumps in the reference's format
 * (src/lbvh/extra/dump.c): which = -1 for the TLAS, otherwise a blas_id.
 * *out_count receives the node count; out_nodes (may be NULL) must hold that many. */
int rthip_get_tree(rthip_ctx* ctx, int which, rthip_node* out_nodes, uint64_t* out_count);

/* ---- ray records (debugging) ---------------------------------------------- */

/* What a BUILD_DEBUG build of the reference appends to ./out.rays for every ray it traces
 * (rt_cpu_dump_add_ray_hit_record / _miss_record, src/raytracer/cpu/extra/dump.c:11-32, called at
 * raytracer_cpu.c:260-262 and :355-357; read by extra/blender.py): the ray, and for a hit the position, shading
 * normal and distance as the intersection routine reports them (before the material turns the normal), for a miss
 * zeros and t = -1.  `material` is the index into the materials array of rthip_tlas_build (-1: miss, or an instance
 * without material); the reference writes the material's handle there. */
typedef struct rthip_ray_record {
    float origin[3];
    float direction[3];
    float p[3];
    float n[3];
    float t;
    int32_t material;
} rthip_ray_record;
/* Renders the frame like rthip_cast (same seed -> same paths) and returns the records of all its rays in the
 * reference's order: pixels row by row, a pixel's samples in sample order, a path's rays from the camera ray on.
 * out_records == NULL: only *out_count is set.  Fails with RTHIP_E_LIMIT when capacity is too small (and *out_count
 * tells the number needed).  A debugging aid: one thread per path, nothing tuned. */
int rthip_cast_ray_records(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, int width, int height,
                           rthip_ray_record* out_records, uint64_t capacity, uint64_t* out_count);

/* ---- mesh files ---------------------------------------------------------- */

/* ms_load_obj, src/mesh/mesh.c:73-219 (the demos load their meshes with it, src/demos/bunny/main.c:44): a Wavefront
 * OBJ file as an indexed triangle list whose vertices are the distinct (position, normal, uv) index triples of the
 * faces, numbered in the order of first appearance (mesh.c:19-37, 183-190), each attribute in a 16-byte slot
 * (src/geo/geo.c:2-18).  attrs: GEO_VertexAttributes bits P = 1, N = 2, T = 4 (0 = all three, mesh.c:85-87); the face
 * syntax must be the one the reference reads for them: "f a b c" (P), "a/t" (PT), "a//n" (PN), "a/t/n" (PNT); of a
 * face with more corners the first three are used (mesh.c:145), and the last line of a file that ends in a newline
 * counts twice, as it does in the reference's read loop (`while (!feof) { fgets ... }`, mesh.c:108-109, 140-141: a
 * trailing face is a duplicated triangle).  Lines of more than 255 characters are not supported (the reference
 * splits them).  The text is parsed on the host's cores (at most 16
 * threads), the deduplication runs on the GPU; the mesh stays in device memory.
 * Differences from the reference: a face that does not match the syntax is an error (RTHIP_E_INVALID with the line
 * number; the reference prints a message and goes on with indices left over from the previous face), and so is an
 * index beyond the file's records (undefined behaviour there); slots and padding floats no attribute writes are 0. */
typedef struct rthip_obj_info {
    uint32_t vertex_count;  /* MS_Mesh.vertices_count */
    uint32_t index_count;   /* MS_Mesh.indices_count  */
    uint32_t attrs;
    uint32_t vertex_bytes;  /* 16 per attribute: geo_vertex_size */
    float parse_ms;         /* host: map, count, parse                          */
    float dedup_ms;         /* device: sort, scans, gather                      */
    float total_ms;         /* the whole call, uploads included                 */
    uint32_t host_threads;
} rthip_obj_info;
int rthip_obj_load(rthip_ctx* ctx, const char* path, uint32_t attrs, rthip_obj_info* out_info);
/* Copies the loaded mesh to the host: MS_Mesh.vertices (vertex_count x vertex_bytes) and MS_Mesh.indices; either may
 * be NULL. */
int rthip_obj_get(rthip_ctx* ctx, float* out_vertices, uint32_t* out_indices);
/* rthip_blas_add_mesh for the loaded mesh, device to device (between rthip_blas_begin and rthip_blas_commit): what
 * rt_world_add_mesh + rt_tracer_build_blas do with the loader's result (src/demos/bunny/main.c:44-52) without the
 * vertices visiting the host. */
int rthip_blas_add_loaded_obj(rthip_ctx* ctx, uint32_t blas_id);

/* ---- image output ------------------------------------------------------- */

/* The reference writes its image with stbi_write_hdr(path, width, height, 3, rgb)
 * (src/third_party/stb/stb_image_write.h:794, called at src/demos/cornell/main.c:222,
 * bunny/main.c:73, spheres/main.c:68, tri/main.c:58): a Radiance header and per scanline
 * the RGBE bytes (stb_image_write.h:639-654), run-length coded per component for
 * 8 <= width < 32768 (:672-759).  rthip_hdr_encode produces exactly those file bytes on
 * the GPU, so that at most ~4 bytes per pixel return to the host and the caller only
 * has to fwrite them.
 *
 * rgb: float[height][width][3] in host memory, or NULL to encode the image of the last
 * rthip_cast, which is still in device memory (no upload; width / height must match it).
 * out: host buffer of `capacity` bytes; *out_size receives the file length.  A capacity
 * of rthip_hdr_bound(width, height) always suffices; with less the call fails with
 * RTHIP_E_LIMIT when the file is longer (and *out_size still tells the length).
 * Components that are negative, NaN or >= 2^31 after scaling convert as the reference's
 * x86-64 build converts them (truncate to int32, low byte; out of range -> 0). */
uint64_t rthip_hdr_bound(int width, int height);
int rthip_hdr_encode(rthip_ctx* ctx, const float* rgb, int width, int height, uint8_t* out, uint64_t capacity,
                     uint64_t* out_size);
/* The same between device buffers, enqueued on `stream` without synchronising: d_rgb as
 * rthip_cast_device writes it, d_out of capacity >= rthip_hdr_bound(width, height),
 * *d_size (device memory, 8 bytes) receives the file length. */
int rthip_hdr_encode_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_out,
                            uint64_t capacity, uint64_t* d_size, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTHIP_H */
