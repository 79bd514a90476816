// Internal declarations shared by the translation units of librthip.so.
// gfx950 (MI355X) only: wave = 64 lanes, no other target is considered.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "rthip.h"

// ---------------------------------------------------------------------------
// device memory that grows on demand and is reused between calls
// ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool fixed = false;  // a slice of memory the library does not own: never freed, never grown (rthip_reserve fails instead)
};

// ---------------------------------------------------------------------------
// LBVH builder workspace (lbvh_build.hip)
// ---------------------------------------------------------------------------
struct ChildInfo;  // 32-byte record a child leaves for its parent

struct LbvhWorkspace {
    DevBuf aabb;        // float[6n]   primitive boxes (when the builder owns them)
    DevBuf keys[2];     // uint64[n]   Morton codes, ping-pong
    DevBuf vals[2];     // uint32[n]   0-based primitive index, ping-pong
    DevBuf block_hist;  // uint32[256 * tiles] per-tile digit counts, digit-major
    DevBuf scan_aux;    // uint32[...] block sums of the device-wide scans
    DevBuf prio;        // uint8[n-1]  split priority of every gap
    DevBuf child;       // ChildRec[2(n-1)]  box climb: records for parents that span chunks
    DevBuf vis;         // per tile of 2048 gaps: what the tile does to the mask of running-minimum records, and the mask behind it
    DevBuf flags;       // uint32[n-1] arrival counters (+ the starter count)
    DevBuf starters;    // gaps whose two children both arrived in the chunk-local kernel (start list of the upper kernel)
    DevBuf lefts;       // uint8[n]    of every sorted leaf: how many of its ancestors hold it in their left subtree (its pre-order index is 2 i + that)
    DevBuf bounds;      // ordered-int centroid bounds, in several copies that the Morton kernel folds
};

// One LBVH in HBM, pre-order, 2 x float4 per node:
//   lo = {min.x, min.y, min.z, bits(skip)}   hi = {max.x, max.y, max.z, bits(id)}
struct DevTree {
    float4* nodes = nullptr;
    uint32_t n_nodes = 0;
};

// Per-mesh data the traversal kernels read.
struct DevMesh {
    uint32_t wide_base;  // first record of this mesh in the fast mode's record array (records of 8 float4)
    uint32_t node_base;  // first node of this BLAS in the shared node array (in nodes, not float4s)
    uint32_t n_nodes;
    uint32_t tri_base;   // first triangle of this mesh in the shared triangle arrays
    uint32_t n_tris;
    uint32_t has_normals;
};

// Per-instance record, 96 bytes, read when a TLAS leaf is entered.
struct DevInstance {
    uint32_t type;
    int32_t material;
    uint32_t node_base;
    uint32_t n_nodes;
    uint32_t tri_base;
    uint32_t has_normals;
    float sphere_radius;
    uint32_t wide_base;  // first record of this mesh in the fast mode's record array (records of 8 float4)
    float sphere_center[4];
    float translation[4];
    float rotation[4];
    float scale[4];
};

struct DevScene {
    const float4* tlas;       // TLAS nodes
    uint32_t n_tlas;          // node count (0: empty world)
    const float4* blas;       // all BLAS nodes, mesh after mesh
    const float4* tri_v;      // 3 x float4 per triangle: v0, v1, v2 (w unused), mesh after mesh
    const float4* tri_n;      // 3 x float4 per triangle: n0, n1, n2 (meshes without normals: unused)
    const float4* tlas_wide;  // RTHIP_TRAV_FAST: the TLAS collapsed to 4-wide records of 8 x float4 (wide_emit_kernel)
    const float4* blas_wide;  // the same for every mesh, mesh after mesh
    const DevInstance* inst;  // by TLAS id - 1
    const rthip_material* mats;
    uint32_t n_inst;
    uint32_t n_mats;
    uint32_t max_bounces;
    uint32_t sky;
};

struct HostMesh {
    DevBuf interleaved;  // raw copy of the caller's vertex buffer
    DevBuf indices;      // uint32[index_count] (absent: auto-index)
    DevBuf px, py, pz;   // SoA positions
    DevBuf nx, ny, nz;   // SoA normals (optional)
    uint32_t vertex_count = 0, stride = 0, pos_off = 0, index_count = 0;
    int32_t nrm_off = -1;
    uint32_t n_tris = 0;
    bool present = false;
    DevMesh dev{};
};

// The mesh of the last rthip_obj_load, in device memory (obj_load.hip)
struct ObjState {
    bool loaded = false;
    uint32_t attrs = 0, vertex_count = 0, index_count = 0;
    DevBuf cp, cn, ct;          // uint32[corners]: 1-based position / normal / uv index of every face corner
    DevBuf pos, nrm, uv;        // the file's `v`, `vn`, `vt` records
    DevBuf head, first, run_first;  // deduplication scratch
    DevBuf indices;             // uint32[corners]
    DevBuf px, py, pz, nx, ny, nz;  // SoA vertices
    DevBuf interleaved;         // the same in the reference's layout, one 16-byte slot per attribute
};

// Every launch of a persistent kernel owns a slot of work counters (1 KiB: eight 128-byte-spaced counters and an arrival
// word) from the moment it is enqueued until its last wave has left.  The first kEagerSlots slots serve ordinary
// launches: a slot is taken only when the event recorded behind its previous launch has completed.  Launches that are
// captured into a HIP graph bake the slot's address into the graph, so each takes one of the remaining slots for good
// (until the scene is rebuilt, which invalidates such graphs anyway).
constexpr uint32_t kEagerSlots = 64;
constexpr uint32_t kCaptureSlots = 192;
constexpr uint32_t kWorkSlots = kEagerSlots + kCaptureSlots;

// Host threads that stay with a context and run the pieces of a job handed to them (rthip_cast's movers: started once,
// so that a call pays a wake-up instead of eight thread starts).
struct HostPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, done;
    std::function<void(uint32_t)> job;
    uint64_t generation = 0;
    uint32_t pieces = 0, next = 0, finished = 0;
    bool stop = false;
    void start(uint32_t n, int device);
    void run(uint32_t n_pieces, std::function<void(uint32_t)> fn);  // returns when every piece has run
    ~HostPool();
};

struct rthip_ctx {
    int device = 0;
    std::string err;
    uint32_t max_bounces = 8;
    int sky = 1;
    int counting = 0;
    int sort_top_bytes = 4;     // large LBVH builds sort by this many upper key bytes and finish locally; 8: all passes (RTHIP_SORT_TOP_BYTES)
    uint64_t sort_top_gave_up_n = 0;  // a build of this many boxes had long stretches left after the upper bytes: all eight passes from then on
    uint32_t* sort_flag_host = nullptr;  // page-locked word: the flag of the last build by the upper bytes, copied behind the build
    hipEvent_t sort_flag_ev = nullptr;   // ... and the event behind that copy
    bool sort_flag_pending = false;
    uint64_t sort_flag_n = 0;            // the number of boxes of the build the pending flag belongs to
    int sort_by_passes = 0;     // RTHIP_SORT_BY_PASSES=1: small sorts also go pass by pass through HBM (diagnostic)
    uint32_t sort_merge_tile = 0;           // keys per tile of the sort-in-LDS-then-merge path: 1024, 2048 or 4096; 0: by size (RTHIP_SORT_MERGE_TILE)
    uint64_t sort_merge_max = 1ull << 17;   // sorts of up to this many keys take that path, at most 128 tiles (RTHIP_SORT_MERGE_MAX)
    uint64_t sort_small_tile_max = 1ull << 18;  // sorts of up to this many keys use 1024-key tiles (RTHIP_SORT_SMALL_TILE_MAX)
    int upper_ranges = 2;       // builds of up to 2^17 leaves take their inner nodes from leaf ranges: 2 all of them, 1 only those above the chunks (generations of hand-offs inside), 0 hand-offs throughout (RTHIP_UPPER_RANGES)
    int upper_device_wide = 0;  // RTHIP_UPPER_DEVICE_WIDE=1: never run the upper climbs in one workgroup (diagnostic)
    int refit_fence = 2;  // hand-offs of the upper hierarchy kernels: 0 release / acquire through the counter, 1 __threadfence(), 2 relaxed + s_waitcnt (default, once checked)
    bool refit_checked = false;  // mode 2 has passed handoff_self_check on this context's device (or RTHIP_REFIT_FENCE chose the mode)

    struct Pinned { void* p; size_t bytes; };
    std::vector<Pinned> pinned;    // page-locked host memory handed out by rthip_host_alloc
    hipStream_t stream = nullptr;  // the context's own stream for host-blocking entry points
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // rthip_cast into page-locked memory: the frame in bands of rows, each on a stream of its own, its copy behind it
    static constexpr int kMaxBands = 8;
    hipStream_t band_stream[kMaxBands] = {};
    hipStream_t band_copy_stream[kMaxBands] = {};
    hipEvent_t band_done[kMaxBands] = {};
    static constexpr int kBandChunks = 4;    // a band's copy into staging goes in this many pieces, each with an event
    hipEvent_t chunk_done[kMaxBands * kBandChunks] = {};
    int cast_bands = 4;  // RTHIP_CAST_BANDS; 1 = one launch, one copy
    void* inst_stage = nullptr;  // page-locked memory in which rthip_tlas_build writes the device's instance records (grown on demand)
    size_t inst_stage_bytes = 0;
    void* host_stage = nullptr;  // page-locked staging for rthip_cast into pageable memory (grown on demand)
    HostPool* movers = nullptr;  // the threads that move staged bands into the caller's buffer (started at first use)
    size_t host_stage_bytes = 0;

    LbvhWorkspace ws;
    LbvhWorkspace ws_static;  // the same arrays carved out of the device's static scratch array (lbvh_static_workspace_init)
    bool ws_static_ready = false;

    std::vector<HostMesh> meshes;
    bool blas_committed = false;
    DevBuf blas_nodes;  // float4[2 * total nodes]
    DevBuf tri_v;       // float4[3 * total tris]
    DevBuf tri_n;       // float4[3 * total tris]
    uint64_t total_blas_nodes = 0, total_tris = 0;

    int traversal = 0;        // RTHIP_TRAV_EXACT / RTHIP_TRAV_FAST
    DevBuf blas_wide;         // float4[8 * (total tris - meshes)]: room for the fast mode's records of every BLAS
    DevBuf tlas_wide;         // float4[8 * (n_inst - 1)]
    DevBuf wide_rank;         // uint32[nodes + 1]: scratch of the conversion
    bool blas_wide_ready = false, tlas_wide_ready = false;
    DevBuf tlas_nodes;  // float4[2 * (2 n_inst - 1)]
    DevBuf inst;        // DevInstance[n_inst]
    DevBuf mats;        // rthip_material[n_mats]
    DevBuf inst_aabb;   // float[6 n_inst]
    uint32_t n_inst = 0, n_mats = 0;
    bool tlas_built = false;

    DevBuf counters;  // uint64[4]
    DevBuf work_ring;  // kWorkSlots x 1 KiB: eight 128-byte-spaced work counters per launch of a persistent kernel
    uint32_t work_next = 0;
    hipEvent_t slot_done[kEagerSlots] = {};  // recorded behind the last launch that used the slot (created on first use)
    bool slot_used[kEagerSlots] = {};
    uint32_t capture_next = 0;               // capture slots handed out since the last scene build
    int cast_mode = 3;           // 0: persistent kernel, 1: fixed pixel per thread, 2: wavefront, 3: 0 or 1 by scene size (RTHIP_CAST=...)
    uint32_t persistent_blocks = 256u * 5u;  // resident workgroups of the persistent kernel (CUs x 5) where the runtime cannot tell
    uint32_t cus = 256u;
    bool persistent_blocks_forced = false;   // RTHIP_PERSISTENT_PER_CU
    struct Resident { const void* kernel; size_t lds; uint32_t blocks; };
    std::vector<Resident> resident;          // workgroups of a persistent kernel the GPU holds at once, per kernel and LDS size
    DevBuf wf[11];               // path-state arrays of the wavefront renderer
    uint64_t wf_capacity = 1ull << 23;  // paths per wavefront batch
    uint32_t inst_batch = 0;     // persistent kernel chosen by name: 0 = enter instances at once, k = park and enter k lanes at a time (RTHIP_INST_BATCH)
    // pixel- / sample-per-lane kernels: instance entries of a wave in batches (entry_batch, trace.hip)
    bool small_meshes = false;        // set by rthip_tlas_build: the mesh instances average at most 32 nodes
    int samples_inst_batch_env = -1;  // RTHIP_SAMPLES_INST_BATCH
    uint32_t rays_inst_batch = 8;     // exact walk, resident ray-stream kernel, scenes of >= 64 instances: the same (RTHIP_RAYS_INST_BATCH)
    uint32_t fast_inst_batch = 8;     // ordered walk: lanes parked at instances enter k at a time, 0 = at once (RTHIP_FAST_INST_BATCH)
    bool batch_known = false;         // the measured decision for batch_key (scene generation, frame size, samples, sharding)
    uint64_t batch_key[4] = {0, 0, 0, 0};
    uint32_t batch_value = 0;
    double probed_entry_ratio = 0.0;  // instance entries per box test of the last kernel-choice probe
    uint32_t refill_lanes = 16;  // lanes with other work that end a traversal phase (RTHIP_REFILL overrides)
    DevBuf scratch_a, scratch_b, scratch_c, scratch_d, scratch_e;  // staging for host <-> device taps
    ObjState obj;
    DevBuf frame;                // float[3 w h]: the image of the last rthip_cast, kept for rthip_hdr_encode(rgb = NULL)
    int frame_w = 0, frame_h = 0;
    DevBuf hdr_stage, hdr_lengths, hdr_offsets, hdr_out, hdr_size;  // .hdr encoder: coded components, their lengths and file offsets
    // set by mesh_prepare_device: the workspace holds the centroid bounds of this box array (consumed by the next build)
    uint32_t box_bounds = 0;     // Morton codes normalised by the boxes' bounds (BUILD_DEBUG builds of the reference) instead of the centroids'
    const float* bounds_of = nullptr;
    uint64_t bounds_n = 0;
    hipStream_t bounds_stream = nullptr;
    float last_cast_ms = 0.f, last_build_ms = 0.f, last_encode_ms = 0.f;
    uint32_t last_cast_kernel = 0, last_cast_refill = 0;
    // measured kernel choice of RTHIP_KERNEL_AUTO for big frames (see tune_cast); scene_generation changes with every
    // build and tracer setting
    uint64_t scene_generation = 0;
    DevBuf tune_counters;
    DevBuf tune_frame;           // framebuffer of the probe when the cast itself accumulates
    uint32_t cast_accumulate = 0;  // set for the duration of rthip_cast_accumulate_device: pixels are added, not stored
    bool tuned = false;
    uint64_t tuned_key[4] = {0, 0, 0, 0};
    int tuned_kernel = 1;
    uint32_t tuned_refill = 16, tuned_inst_batch = 0;
};

// ---- helpers (rthip_api.hip) -------------------------------------------------
int rthip_fail(rthip_ctx* ctx, int code, const char* fmt, ...);
int rthip_reserve(rthip_ctx* ctx, DevBuf& b, size_t bytes);
void rthip_release(DevBuf& b);

#define RT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rthip_fail((ctx), RTHIP_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                                \
    } while (0)

#define RT_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != RTHIP_OK) return rc_; \
    } while (0)

// ---- LBVH builder (lbvh_build.hip) ---------------------------------------------
// Builds the pre-order tree of n boxes (d_aabb6: n x 6 floats in HBM) into d_nodes
// (2n-1 nodes, 2 float4 each).  Enqueues on `stream`; no host synchronisation.
// When d_codes_unsorted / d_keys_sorted / d_ids_sorted are non-null they receive
// copies of the intermediate arrays (parity taps).  `paired`: records in the walk's own form
// (min.x, max.x, min.y, max.y | min.z, max.z, skip, id) instead of the documented rthip_node; the
// context's BLAS and TLAS are kept that way, what leaves through the ABI is rthip_node.
// `ws`: the workspace to use (nullptr: the context's own, grown on demand).
int lbvh_build_device(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, float4* d_nodes, hipStream_t stream,
                      uint64_t* d_codes_unsorted, uint64_t* d_keys_sorted, uint32_t* d_ids_sorted, bool paired = false,
                      LbvhWorkspace* ws = nullptr);

// Grows the context's workspace to what a build of n boxes needs (lbvh_build_device does so itself on demand).
int lbvh_reserve_workspace(rthip_ctx* ctx, uint64_t n, LbvhWorkspace* ws = nullptr);

// Builds of at most kStaticBuildMaxBoxes boxes can run in a workspace that is a file-scope device array instead of
// allocated memory.  lbvh_static_workspace_init looks the array's address up on the context's device (once per device)
// and carves ctx->ws_static out of it; no allocation.
constexpr uint64_t kStaticBuildMaxBoxes = 1ull << 17;
int lbvh_static_workspace_init(rthip_ctx* ctx);

// Stable LSD radix sort of the n (uint64 key, uint32 value) pairs in ws.keys[*cur] / ws.vals[*cur] (the caller has
// reserved both ping-pong pairs) by the key's low n_bytes bytes; *cur names the pair that holds the result.
// identity_vals: the values are the positions 0 .. n-1 and need not be stored before the call (vals[*cur] is not read).
// first_byte > 0: only the passes over bytes first_byte .. n_bytes-1 (a stable sort by those bytes alone).
// codes_of: the keys are not in keys[*cur] yet; they are the Morton codes of these boxes and the sort makes them itself
// (eight-byte keys and identity values only: see lbvh_build_device), in its tile sort or in its first histogram pass.
struct MortonSource {
    const float* aabb6 = nullptr;      // the boxes
    const uint32_t* bounds = nullptr;  // the centroid bounds' slots
    uint32_t* zero_word = nullptr;     // a word to clear on the way
    uint32_t* zero_word2 = nullptr;    // another (sorts pass by pass only)
    uint32_t* zero_all = nullptr;      // one word per key to clear (sorts pass by pass only)
};
int radix_sort_workspace_device(rthip_ctx* ctx, uint64_t n, int n_bytes, hipStream_t stream, int* cur, LbvhWorkspace* ws = nullptr,
                                bool identity_vals = false, int first_byte = 0, const MortonSource* codes_of = nullptr);
// In-place inclusive prefix sum of n uint32.
int scan_inclusive_u32_device(rthip_ctx* ctx, uint32_t* d, uint64_t n, hipStream_t stream);

// Triangle boxes (+ gathered per-triangle vertex / normal records) from SoA vertex data.
int mesh_prepare_device(rthip_ctx* ctx, const float* px, const float* py, const float* pz, const float* nx,
                        const float* ny, const float* nz, const uint32_t* indices, uint64_t n_tris,
                        float* d_aabb6, float4* d_tri_v, float4* d_tri_n, hipStream_t stream);

int deinterleave_device(rthip_ctx* ctx, const void* d_interleaved, uint32_t vertex_count, uint32_t stride,
                        uint32_t off, float* x, float* y, float* z, hipStream_t stream);

// The fast mode's records (both children's boxes and references per inner node, in pre-order among the inner nodes)
// from a finished pre-order tree in the walk's own record form.  d_wide holds 4 float4 per inner node ((n_nodes - 1) / 2).
int wide_convert_device(rthip_ctx* ctx, const float4* d_nodes, uint32_t n_nodes, float4* d_wide, hipStream_t stream);

int instance_aabbs_device(rthip_ctx* ctx, const DevInstance* d_inst, uint32_t n, const float4* d_blas,
                          float* d_aabb6, hipStream_t stream);

// ---- tracing (trace.hip) ----------------------------------------------------------
// Work-counter slot for one launch of a persistent kernel on `stream` (see kEagerSlots); RTHIP_E_LIMIT when every slot
// is held by an unfinished launch.  work_slot_launched records the launch behind the slot (not for captured launches).
int work_slot_take(rthip_ctx* ctx, hipStream_t stream, uint32_t** out_work, int* out_slot);
int work_slot_launched(rthip_ctx* ctx, int slot, hipStream_t stream);
int trace_rays_launch(rthip_ctx* ctx, const DevScene& scene, const float* d_o, const float* d_d, uint64_t n,
                      uint32_t* d_hits4, float* d_uv, float* d_normal, uint64_t* d_counters, hipStream_t stream);
int occlusion_launch(rthip_ctx* ctx, const DevScene& scene, const float* d_o, const float* d_d, uint64_t n, float tmax,
                     uint8_t* d_out, uint64_t* d_counters, hipStream_t stream);
int primary_hits_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, int width, int height,
                        uint32_t* d_hits4, uint64_t* d_counters, hipStream_t stream);
int cast_wavefront_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed,
                          float* d_rgb, int width, int height, rthip_shard shard, uint64_t* d_counters,
                          hipStream_t stream);
// ray records of a frame: d_slots holds paths x max_bounces records, d_counts paths uint32 (becomes their inclusive
// scan); with d_out the records are compacted into it in path order; *out_total = number of records.  Synchronises.
int cast_records_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, int width,
                        int height, rthip_ray_record* d_slots, uint32_t* d_counts, rthip_ray_record* d_out,
                        uint64_t* out_total, hipStream_t stream);
int cast_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb,
                int width, int height, rthip_shard shard, uint64_t* d_counters, hipStream_t stream);

// ---- .hdr output (hdr_encode.hip) ---------------------------------------------------
uint64_t hdr_bound(int width, int height);
int hdr_encode_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_out, uint64_t capacity,
                      uint64_t* d_size, hipStream_t stream);

// ---- OBJ loading (obj_load.hip) ------------------------------------------------------
int obj_load(rthip_ctx* ctx, const char* path, uint32_t attrs, rthip_obj_info* info);
int obj_read_back(rthip_ctx* ctx, float* out_vertices, uint32_t* out_indices);
