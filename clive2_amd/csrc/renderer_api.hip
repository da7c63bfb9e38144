// renderer_api.hip -- host side of libclive2_amd.so: device-memory ownership, scene repacking,
// launch sequencing on one HIP stream, HIP-event stage timers, and the extern "C" entry points
// declared in include/clive2_amd.h.
//
// Stage order and buffer roles follow Renderer.run_sample (src/renderer.py:281-291); what differs
// is where the state lives (SoA on the device, accumulators included) and that nothing returns to
// the host between samples.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/clive2_amd.h"
#include "kernels.hpp"
#include "tonemap.hpp"
#include "det_splat.hpp"
#include "bvh_builder.hpp"
#include "comm_rccl.hpp"
#include "comm_wait.hpp"

using namespace cl2;

namespace {

thread_local std::string g_create_error;

// Reference AoS records as the host hands them over (src/struct_types.py).
struct BoxRec { float min[4], max[4]; int32_t left, right, pad[2]; };
struct TriRec { float v0[4], v1[4], v2[4], n0[4], n1[4], n2[4], normal[4]; int32_t material, is_light, is_camera, pad; };
struct MatRec { float color[4], emission[4]; int32_t type; float alpha, ior; int32_t pad; };
static_assert(sizeof(BoxRec) == 48 && sizeof(TriRec) == 128 && sizeof(MatRec) == 48 && sizeof(CameraRec) == 112, "ABI");

enum Stage { ST_GENERATE, ST_TRAVERSE_PATHS, ST_BOUNCE, ST_CONNECT_SETUP, ST_TRAVERSE_CONN, ST_CONNECT_RESOLVE,
             ST_FINALIZE, ST_ACCUMULATE, ST_COUNT };

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
};

}  // namespace

struct cl2_renderer {
    // B = ENTRIES of every per-pixel array = streams x FB (cl2_set_sample_streams; kernels.hpp header): what every launch covers
    // and every stride is; FB = W x H, the pixels of the frame: accumulators, tone map, the multi-GPU reduce, exports of one stream
    int device = 0, W = 0, H = 0, B = 0, FB = 0;
    int streams = 1;                     // independent samples of the frame per pass (one seed buffer each, as K Renderers would hold)
    int export_stream = 0;               // the stream the debug exports / sample-image calls address
    hipStream_t stream = nullptr;        // subpath phase (and everything else when not pipelining)
    hipStream_t stream_conn = nullptr;   // sample pipeline: connection set-up + connection rays
    hipStream_t stream_res = nullptr;    // sample pipeline: resolve + K6 + accumulation
    hipEvent_t ev_paths[3] = {}, ev_conn[2] = {}, ev_res[6] = {};
    hipEvent_t ev_tune[2] = {};          // share tuner: end of the first / of the last sample of a candidate's run (timing events)
    double tune_period_ms = 0.0;         // ... and what they measured: ms per sample between the two
    bool pipe_active = false;            // inside a pipelined cl2_run_samples
    int paths_share = 0;                 // eighths of the wave slots given to the subpath stage while pipelining (0 = not tuned yet)
    int pipelining = -1;                 // sample pipeline inside cl2_run_samples: 0 serial, 1 two stages, 2 three stages, -1 by frame size
    std::string err;
    bool scene_ok = false;
    int counting = 0;                    // 0 off, 1 the reference walk's node / triangle tallies (binary walk), 2 the 4-wide walk's own tallies
    int profiling = 0;                   // 0 off, 1 the connection-ray traversal launch only, 2 every stage
    int debug_flags = 0;
    int traversal_order = 0;             // 0 = the reference's child order (exact); 1 = nearest child first in the 4-wide walks (opt-in, NOT bit-exact: bvh_wide.hpp ORDER)
    int gather_lanes = 32, gather_wait = 48;   // whole-subpath launch: lanes gathered / steps waited before a wave runs its bounce phase
    int traversal_mode = 0;              // 0 auto, 1 fused (one ray per lane), 2 split (persistent traversal + ray replacement)
    unsigned* d_work = nullptr;          // [8][WORK_STRIDE] work counters of the persistent traversal launches, a 64-byte line per launch slot
    int levels_per_launch = 0;           // subpath levels per launch (6 = one launch, 1 = compaction after every bounce, 0 = by survival)
    int levels_auto = 0;                 // the choice made for levels_per_launch == 0 (0 = not made yet)

    // scene
    BvhView bvh{};
    float4 *d_nodes = nullptr, *d_tris = nullptr, *d_tri_shade = nullptr, *d_light_tris = nullptr;
    int* d_light_tri_index = nullptr;
    float* d_light_areas = nullptr;
    MaterialDev* d_mats = nullptr;
    int n_mats = 0, light_count = 0;
    int n_top = 0;                       // boxes of the top levels renumbered to the front of the record array (0: plain visit order)
    // 4-wide collapse of the tree for the exact wide walk (bvh_wide.hpp); n_wide == 0: not available for this scene
    float4* d_wide = nullptr;
    int n_wide = 0;
    int* d_tri_rank = nullptr;           // ORDER: each triangle's position in the reference's visit order (exact-t ties; bvh_wide.hpp)
    float* d_tris36 = nullptr;           // 36-byte triangle records of a tree that streams from beyond L2 (bvh_wide.hpp, PACK); nullptr: none
    int n_fast = 0;                      // records of the pruned table (bvh.n_fast_nodes unless debug_flags bit 7 switches it off)
    CamTris cam_tris{0, {0, 0, 0, 0}};   // the triangles with is_camera set, as kernel arguments of the resolve stage (n < 0: too many, look them up)
    int fast_flat = 0;                   // the pruned table is a plain list of leaves (bvh.fast_flat unless debug_flags bit 11 switches it off)
    float4* d_fast = nullptr;            // pruned record table of an LDS-resident tree (cl2_upload_scene); bvh.n_fast_nodes == 0: none
    WideView wide{};
    int2* d_wide_ovf = nullptr;          // per-lane stack overflow of the wide launches (one region per stage: [2]); allocated by the first wide launch
    int wide_ovf_entries = 0;            // entries per lane: the deepest stack this tree can produce (cl2_upload_scene)
    CameraRec cam{};

    // state
    uint2* d_seeds = nullptr;
    PathBufs sets[3][2]{};             // subpath buffer sets of the sample pipeline
    int cur = 0;                       // the set stage calls and exports use (after run_samples: the last sample's)
    float4* d_hit = nullptr;
    float4* d_hit_cam0 = nullptr;      // closest hits of the camera subpaths' first rays, traced together with the light subpaths' (launch_trace)
    int* d_queue = nullptr;            // [6][B], shared by both subpath kinds (they run one after the other)
    unsigned* d_qcount = nullptr;      // [9]: [0] = B (level-0 count), [1..6] level counts, [7] connection rays, [8] = 2B
    int* d_ctag = nullptr;             // connection-ray queue: {slot, pixel} tags
    // reproducible light image (cl2_set_reproducible, det_splat.hpp): records of one pass, allocated on first use
    bool reproducible = false;
    unsigned *d_det_keys = nullptr, *d_det_keys_sorted = nullptr, *d_det_slots = nullptr, *d_det_slots_sorted = nullptr;
    float4* d_det_vals = nullptr;
    void* d_det_tmp = nullptr;
    size_t det_tmp_bytes = 0;
    float2* d_chit[2] = {nullptr, nullptr};            // two sets: connection rays of sample i+1 vs resolve of sample i
    unsigned long long* d_cmask[2] = {nullptr, nullptr};
    float* d_agg = nullptr;
    float4 *d_light_image = nullptr, *d_finalized = nullptr, *d_uni = nullptr;
    float* d_sample_w = nullptr;
    float* d_acc = nullptr;            // [8][B]
    double* d_tone_partial = nullptr;  // device tone map: per-workgroup partial sums + the total (allocated by the first call)
    uint8_t* d_tone_out = nullptr;     // device tone map: the uint8 picture before it is copied out
    Stats* d_stats = nullptr;
    unsigned long long* d_block_stats = nullptr;   // [grid][4]: rays, box tests, tri tests, counted rays per workgroup slot

    // host-side tallies
    uint64_t samples = 0, counted_rays = 0;
    double ms[ST_COUNT] = {0};
    uint64_t launches_tp = 0, launches_tc = 0, rays_tp = 0, rays_tc = 0;
    struct Span { hipEvent_t a, b; int stage; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;

    std::vector<void*> allocs;

    // multi-GPU sample split: one RCCL communicator per handle (cl2_comm_init_rank)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_nranks = 0;
    bool comm_poisoned = false;          // a collective failed or timed out (or the host said so): tear down with ncclCommAbort
    double* d_comm_scratch = nullptr;    // [COMM_SCRATCH] doubles for cl2_comm_allreduce_f64
};

namespace {

#define HIP_TRY(r, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (r)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return CL2_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)

#define TRY(expr)                         \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != CL2_OK) return rc_;    \
    } while (0)

template <typename T> int dev_alloc(cl2_renderer* r, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { r->err = std::string("hipMalloc: ") + hipGetErrorString(e); return CL2_E_NOMEM; }
    r->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return CL2_OK;
}

template <typename T> void dev_free(cl2_renderer* r, T*& p) {
    if (!p) return;
    auto it = std::find(r->allocs.begin(), r->allocs.end(), static_cast<void*>(p));
    if (it != r->allocs.end()) r->allocs.erase(it);
    (void)hipFree(p);
    p = nullptr;
}

int fail(cl2_renderer* r, int code, const std::string& msg) { r->err = msg; return code; }
inline float as_f_int(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

inline int grid_for(size_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

// nullptr when no event can be created: the span is then simply not timed (profiling is best effort,
// the launch itself is unaffected)
hipEvent_t take_event(cl2_renderer* r) {
    if (!r->event_pool.empty()) { hipEvent_t e = r->event_pool.back(); r->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Timed {   // records a HIP-event span around a launch when profiling is on
    cl2_renderer* r; int stage; hipStream_t st; hipEvent_t a = nullptr;
    Timed(cl2_renderer* r_, int stage_, hipStream_t st_) : r(r_), stage(stage_), st(st_) {
        // level 1: the two traversal stages only (the connection-ray launch and the subpath traversal launches: bench.py's rooflines)
        const bool wanted = r->profiling >= 2 || (r->profiling == 1 && (stage == ST_TRAVERSE_CONN || stage == ST_TRAVERSE_PATHS));
        if (wanted) {
            a = take_event(r);
            if (a && hipEventRecord(a, st) != hipSuccess) { r->event_pool.push_back(a); a = nullptr; }
        }
    }
    ~Timed() {
        if (!a) return;
        hipEvent_t b = take_event(r);
        if (b && hipEventRecord(b, st) == hipSuccess) { r->spans.push_back({a, b, stage}); return; }
        r->event_pool.push_back(a);
        if (b) r->event_pool.push_back(b);
    }
};

int drain(cl2_renderer* r) {
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream_conn));
    HIP_TRY(r, hipStreamSynchronize(r->stream_res));
    for (auto& s : r->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) r->ms[s.stage] += ms;
        r->event_pool.push_back(s.a);
        r->event_pool.push_back(s.b);
    }
    r->spans.clear();
    return CL2_OK;
}

// ---- launches (asynchronous on the given stream; `set` = the two subpath buffers of the sample) ----
int launch_generate(cl2_renderer* r, int which, hipStream_t st, const PathBufs* set) {
    Timed t(r, ST_GENERATE, st);
    if (which == CL2_LIGHT)
        hipLaunchKernelGGL(k_gen_light_rays, dim3(grid_for(r->B)), dim3(BLOCK), 0, st, r->B, r->d_light_tris,
                           r->d_light_areas, r->d_light_tri_index, r->d_mats, r->light_count, r->d_seeds, set[CL2_LIGHT]);
    else
        hipLaunchKernelGGL(k_gen_camera_rays, dim3(grid_for(r->B)), dim3(BLOCK), 0, st, r->B, r->cam, r->d_seeds,
                           set[CL2_CAMERA]);
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

int launch_generate_both(cl2_renderer* r, hipStream_t st, const PathBufs* set) {
    Timed t(r, ST_GENERATE, st);
    hipLaunchKernelGGL(k_gen_rays, dim3(grid_for(r->B)), dim3(BLOCK), 0, st, r->B, r->d_light_tris, r->d_light_areas,
                       r->d_light_tri_index, r->d_mats, r->light_count, r->cam, r->d_seeds, set[CL2_LIGHT], set[CL2_CAMERA]);
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

// Trees that are not LDS-resident.  Connection rays (one big launch of rays of very different cost)
// run as a persistent launch with lane-level ray replacement.  Subpaths have two organisations:
//   persistent (mode 2, the automatic choice)  one persistent traversal launch + one bounce launch per
//               level: best lane use, but 24 small launches per sample, each ending in a tail;
//   fused (mode 3)  the whole-subpath kernel of the small scenes, its tree fetched through the caches:
//               one ray per lane (a wave is as slow as its slowest ray), but 2 launches and no tails.
// In serial order fused wins on mid-size trees (5k-triangle sphere: 5.7 vs 8.7 ms per sample), but
// the sample pipeline with stage shares hides the tails of the persistent form and then the two are
// equal there (6.05 vs 6.00 Grays/s) and persistent wins from 82k triangles on (4.7 vs 3.9).
// dynamic shared memory of a launch that stages the tree (stage_bvh)
inline size_t bvh_lds_bytes(const cl2_renderer* r) {
    return ((size_t)2 * r->bvh.n_lds_nodes + (r->bvh.lds_tris ? (size_t)3 * r->bvh.n_tris : 0) + (size_t)2 * r->bvh.n_fast_nodes) * sizeof(float4);
}
// counting mode 1: the node / triangle tallies of the REFERENCE's walk (binary stackless walk; the wide walk steps aside);
// mode 2 tallies what the 4-wide walk itself fetches and changes no launch but the whole-subpath one (see whole_subpaths)
inline bool count_ref(const cl2_renderer* r) { return r->counting == 1; }
inline bool tree_in_lds(const cl2_renderer* r) { return r->bvh.lds_tris && r->bvh.n_nodes <= r->bvh.n_lds_nodes; }
inline bool split_paths(const cl2_renderer* r) {
    if (r->traversal_mode == 1 || r->traversal_mode == 3) return false;
    if (r->traversal_mode == 2 || r->traversal_mode >= 4) return true;
    return !tree_in_lds(r);
}
inline bool split_conn(const cl2_renderer* r) {
    return r->traversal_mode >= 2 || (r->traversal_mode == 0 && !tree_in_lds(r));
}
// The exact 4-wide walk (bvh_wide.hpp) for the connection-ray launch and, while the sample pipeline runs, the per-level subpath
// launches: mode 5, and the automatic choice for every tree that is read through the caches.  Round 2 kept the binary walk above
// 16 MB (15.7 vs 18.6 ms on the 155 MB tree, one triangle per pass); with two triangle pairs per pass, 7 stack entries in LDS and
// the gathered refill the wide walk is ahead there too (round 3, same box: connection launch 14.08 -> 13.51 ms, sample 27.78 ->
// 27.22 ms).  Never while counting: the node-test tallies are defined by the binary walk.
inline bool wide_walk(const cl2_renderer* r) {
    if (r->n_wide <= 0 || count_ref(r)) return false;
    if (r->traversal_mode == 5) return true;
    return (r->traversal_mode == 0 || r->traversal_mode == 3 || r->traversal_mode == 4) && !tree_in_lds(r);
}
// Whole subpaths (light, then camera, all levels) in ONE persistent launch per sample (k_subpaths_persistent)
// instead of a traversal + a bounce launch per level and kind: mode 4.  Measured at 1080p (ms per sample, serial order /
// sample pipeline): glass 15.9 / 12.6 per level, 13.0 / 13.6 whole; interior 36.2 / 32.7 per level, 33.5 / 34.1 whole.
// One launch has one tail instead of 24, which is what the serial order pays for; the sample pipeline already fills
// those tails with the other stage's work, and there the whole-subpath launch loses to its own costs (5 instead of 8
// waves per SIMD for the bounce code's registers; vertex stores scattered over unrelated pixels: 7.7 GB written per
// launch against 1.7 GB of vertices).  And a level launch of a 4K frame (8.3 M rays, 16 per lane) has little tail to
// lose: there the per-level form wins in serial order too (interior 4K: 53.7 vs 61.6 ms of subpath time, blob 24.0 vs
// 28.1).  So the automatic choice takes it in the serial order and up to 2^22 pixels only.
inline bool whole_subpaths(const cl2_renderer* r) {
    if (r->traversal_order != 0) return false;                       // nearest-first order exists in k_traverse_wide only: every ray goes through it
    if (r->counting == 2 && r->traversal_mode != 4) return false;   // the walk's own tallies are taken in k_traverse_wide: every ray goes through it
    return r->traversal_mode == 4 || (r->traversal_mode == 0 && !tree_in_lds(r) && !r->pipe_active && r->B <= (1 << 22));
}
// Two triangles per step of the persistent walk while the tree is cache-resident (the step is then
// issue-bound and fewer, fatter steps win: glass +5 %, blob +4 %); one when it streams from memory
// (1M triangles: two cost 3 %).  debug_flags bit 12 inverts the choice (tests run both forms).
inline bool two_tris_per_step_plain(const cl2_renderer* r) {      // the size rule itself: the tree is cache-resident (<= 16 MB)
    return (size_t)r->bvh.n_nodes * 32 + (size_t)r->bvh.n_tris * 48 <= ((size_t)16 << 20);
}
inline bool two_tris_per_step(const cl2_renderer* r) {
    const bool two = two_tris_per_step_plain(r);
    return ((r->debug_flags >> 12) & 1) ? !two : two;
}
inline int effective_levels(const cl2_renderer* r) {
    if (r->levels_per_launch > 0) return std::min(r->levels_per_launch, (int)MAX_VERTS);
    return r->levels_auto ? r->levels_auto : (int)MAX_VERTS;
}
// Stages of the sample pipeline.  Small frames do not fill the machine with one launch (256x256: 256
// workgroups on 256 CUs), so a third stage side by side pays (10.1 / 13.9 / 18.4 Grays/s with 0 / 1 / 2);
// from 1080p on two and three stages measure the same and two need less memory traffic in flight.
inline int pipeline_stages(const cl2_renderer* r) {
    if (r->pipelining >= 0) return r->pipelining;
    return r->B <= (1 << 19) ? 2 : 1;
}
inline int persistent_grid() { return 256 * 8; }      // 256 CUs x 8 workgroups of 4 waves = 32 waves per CU
// While the sample pipeline runs, the subpath stage and the connection stage of two samples are on the
// machine together.  Persistent launches hold their wave slots until the launch runs dry, so the two
// stages get a fixed share of the slots each (in eighths of the machine): the stage whose launch is in
// its tail (a few long rays) leaves the VALUs to the other one instead of leaving them idle.
constexpr int TUNE_SAMPLES = 6;       // samples timed per candidate organisation
constexpr int SHARE_SERIAL = 9;       // tuned result: the serial order beats every pipelined organisation
inline int paths_eighths(const cl2_renderer* r) {
    const int forced = (r->debug_flags >> 8) & 7;          // experiment switch
    return forced ? forced : ((r->paths_share && r->paths_share != SHARE_SERIAL) ? r->paths_share : 4);
}
// share 8 = no fixed shares: both stages launch full-size grids and take the slots as they come
// (large frames: the launches are long, their tails do not matter, and a full grid hides latency best)
inline int persistent_grid_paths(const cl2_renderer* r) {
    const int e = paths_eighths(r);
    return (r->pipe_active && e < 8) ? 256 * e : persistent_grid();
}
inline int persistent_grid_conn(const cl2_renderer* r) {
    const int e = paths_eighths(r);
    return (r->pipe_active && e < 8) ? 256 * (8 - e) : persistent_grid();
}

// One persistent launch of the exact 4-wide walk.  `stage` 0 = subpath stage, 1 = connection stage: each has its own
// stack-overflow region, since the two run side by side in the sample pipeline.
// The per-lane stack of a wide walk keeps its first entries in LDS and the rest in a global array.  How deep it can get
// is a static property of the tree: every entry pending on the reference's stack (at most `pending` of them, counted at
// upload) stands for at most two wide-stack entries, plus the four a visit pushes.  The array is sized for that bound
// (not for the 2 x 64 + 8 any admissible tree could reach) and exists only once a wide launch has been asked for:
// an LDS-resident scene never takes the wide walk on its own, and a turntable builds one renderer per frame.
int ensure_wide_overflow(cl2_renderer* r) {
    if (r->d_wide_ovf) return CL2_OK;
    const size_t lanes = (size_t)persistent_grid() * BLOCK;
    return dev_alloc(r, &r->d_wide_ovf, 2 * lanes * (size_t)std::max(r->wide_ovf_entries, 1));
}

template <class Source>
int launch_wide(cl2_renderer* r, hipStream_t st, int stage, const unsigned* count, unsigned* work_counter, Source src, int is_conn) {
    const int grid = stage == 0 ? persistent_grid_paths(r) : persistent_grid_conn(r);
    TRY(ensure_wide_overflow(r));
    WideView w = r->wide;
    w.ovf_stride = std::max(r->wide_ovf_entries, 1);
    w.overflow = r->d_wide_ovf + (size_t)stage * persistent_grid() * BLOCK * w.ovf_stride;
    // LDS per workgroup: per-lane stack (8 B per entry and lane) + the top of the tree (128 B per wide node); experiment
    // switches: debug_flags bits 16-19 stack entries (0 = default), bits 20-23 window in units of 32 wide nodes
    const int sflag = (r->debug_flags >> 16) & 0xF, wflag = (r->debug_flags >> 20) & 0xF;
    // (round 3 ran 7 stack entries + a 64-node window, 22 KB per workgroup; its measurements are in DESIGN.md 6.1)
    // round 4: WIDE_STACK_LDS (8) entries, a compile-time constant (shift-addressed).  The window's lanes read LDS in a
    // branch of their own (the per-lane pointer select of round 3 made EVERY node fetch a flat load).  Same-box A/B of the
    // rewritten walk, ms (connection launch alone | sample), stack entries + window nodes:
    //   glass     8+0 3.28 | 8.17   8+32 3.09 | 7.90   7+64 3.19 | 7.79   8+64 3.35 | 7.98   6+64 3.11 | 7.91   6+128 3.68 | 8.68
    //   blob      8+0 4.18 | 10.30  8+32 3.90 | 9.95   7+64 4.02 | 9.97   8+64 4.15 | 10.15
    //   interior  8+0 12.0 | 24.2   8+32 11.2 | 22.8   7+64 11.2 | 23.3   8+64 11.2 | 22.3; one triangle pair per pass: 8+64 10.7 | 21.3
    // so: 32 nodes (20 KB per workgroup: 8 workgroups per CU) and two pairs per pass while the tree is cache-resident, 64
    // nodes and one pair when it streams from memory.  debug_flags bits 20-23 override the window (units of 32 nodes, 15 = none).
    (void)sflag;
    const bool streams_from_memory = !two_tris_per_step_plain(r);
    w.stack_lds = WIDE_STACK_LDS;
    w.n_lds_nodes = std::min(r->n_wide, wflag == 15 ? 0 : (wflag ? 32 * wflag : (streams_from_memory ? 64 : 32)));
    // never more than the 64 KB a workgroup may ask for: the window gives way, the stack entries are needed
    w.n_lds_nodes = std::min<int>(w.n_lds_nodes, (int)(((size_t)64 * 1024 - (size_t)w.stack_lds * BLOCK * 8) / 128));
    const size_t lds = (size_t)w.stack_lds * BLOCK * 8 + (size_t)w.n_lds_nodes * 128;
    // the binary records (lanes whose ray has a non-finite 1/d) come through the caches: no window for them
    BvhView b = r->bvh;
    b.n_lds_nodes = 0; b.lds_tris = 0; b.n_fast_nodes = 0;
#define CL2_WIDE(REPS, TALLY, SPEC, PACK, ORDER) \
    hipLaunchKernelGGL((k_traverse_wide<REPS, Source, TALLY, SPEC, PACK, ORDER>), dim3(grid), dim3(BLOCK), lds, st, w, b, count, work_counter, src, r->d_stats, is_conn)
// (a macro argument may not be a run-time value: one dispatch level per template parameter)
#define CL2_WIDE_BY_SPEC(REPS, TALLY, PACK) do { if (spec) CL2_WIDE(REPS, TALLY, true, PACK, false); else CL2_WIDE(REPS, TALLY, false, PACK, false); } while (0)
#define CL2_WIDE_BY_TALLY(REPS, PACK) do { \
        if (r->traversal_order != 0) { if (r->counting == 2) CL2_WIDE(REPS, true, true, PACK, true); else CL2_WIDE(REPS, false, true, PACK, true); } \
        else if (r->counting == 2) CL2_WIDE_BY_SPEC(REPS, true, PACK); else CL2_WIDE_BY_SPEC(REPS, false, PACK); } while (0)
    const bool spec = !((r->debug_flags >> 13) & 1);            // speculative expansion of the stack top (bvh_wide.hpp); bit 13: off
    // 36-byte triangle records, a pair fetched as one run of 72 bytes (bvh_wide.hpp: PACK); bit 14: the 48-byte records of the other
    // walks.  The opt-in nearest-first child order (cl2_set_traversal_order; NOT the parity path) exists for the speculative walk only.
    const bool pack = w.tris36 && !((r->debug_flags >> 14) & 1);
    if (streams_from_memory) { if (pack) CL2_WIDE_BY_TALLY(1, true); else CL2_WIDE_BY_TALLY(1, false); }
    else { if (pack) CL2_WIDE_BY_TALLY(WIDE_TRI_REPS, true); else CL2_WIDE_BY_TALLY(WIDE_TRI_REPS, false); }
#undef CL2_WIDE_BY_TALLY
#undef CL2_WIDE_BY_SPEC
#undef CL2_WIDE
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

// Subpath phase scratch (d_queue, d_qcount[0..6], d_work[0..6], d_hit, d_block_stats) is touched by this
// phase only; the connection phase owns d_qcount[7] and d_work[7].
// `cam0`: 0 = plain; 1 (light subpaths) = the level-0 traversal launch also walks the camera subpaths' first rays (their
// hits go to d_hit_cam0); 2 (camera subpaths) = level 0 has no traversal launch of its own, its bounce reads d_hit_cam0.
// Only the persistent per-level organisation has separate traversal launches to merge.
int launch_trace(cl2_renderer* r, int which, hipStream_t st, const PathBufs* set, int cam0 = 0) {
    const int B = r->B;
    PathBufs pb = set[which];
    HIP_TRY(r, hipMemsetAsync(r->d_qcount + 1, 0, 6 * sizeof(unsigned), st));
    const bool split = split_paths(r);
    if (!split) cam0 = 0;
    if (split) HIP_TRY(r, hipMemsetAsync(r->d_work, 0, 7 * WORK_STRIDE * sizeof(unsigned), st));
    const int step = split ? 1 : effective_levels(r);
    for (int first = 0; first < MAX_VERTS; first += step) {
        const int end = std::min(first + step, (int)MAX_VERTS);
        // queue slot k holds the paths alive after level k-1 (slot 0 = everybody, count B)
        const int* q_in = first == 0 ? nullptr : r->d_queue + (size_t)(first - 1) * B;
        const unsigned* c_in = r->d_qcount + first;
        int* q_out = r->d_queue + (size_t)(end - 1) * B;
        unsigned* c_out = r->d_qcount + end;
        const bool merged = first == 0 && cam0 == 1, hits_ready = first == 0 && cam0 == 2;
        if (split && !hits_ready) {
            Timed t(r, ST_TRAVERSE_PATHS, st);
            PathRaySource src{q_in, pb.P0 + (size_t)first * B, pb.P1 + (size_t)first * B, r->d_hit};
            const PathBufs& cpb = set[CL2_CAMERA];
            DualPathRaySource dual{pb.P0, pb.P1, cpb.P0, cpb.P1, r->d_hit, r->d_hit_cam0, B};
            const unsigned* c_trav = merged ? r->d_qcount + 8 : c_in;
            // The per-level subpath launches take the 4-wide walk wherever the scene has it.  Rounds 2 / 3 took it only while the
            // sample pipeline ran (alone these launches were tail-bound and gained nothing: glass 7.43 -> 7.56 ms); with round 4's
            // pass the walk is ahead alone as well -- serial order at 3840 x 2160, subpath traversal per sample: blob 15.8 -> 12.6
            // ms, 1M triangles 44.2 -> 31.8 -- and with sample streams a launch is no longer all tail.  (debug_flags bit 3, which
            // forced it in the serial order, is accepted and has no effect any more; mode 2 still means the binary walk.)
            if (wide_walk(r)) {
                if (merged) TRY(launch_wide(r, st, 0, c_trav, r->d_work + first * WORK_STRIDE, dual, 0));
                else TRY(launch_wide(r, st, 0, c_trav, r->d_work + first * WORK_STRIDE, src, 0));
                r->launches_tp++;
            } else {
#define CL2_PERSIST(CNT, TWO, SRCT, SRC)                                                                                   \
            hipLaunchKernelGGL((k_traverse_persistent<CNT, TWO, SRCT>), dim3(persistent_grid_paths(r)), dim3(BLOCK), bvh_lds_bytes(r), \
                               st, r->bvh, c_trav, r->d_work + first * WORK_STRIDE, SRC, r->d_stats, 0)
#define CL2_PERSIST_SRC(SRCT, SRC)                                                                                         \
            do {                                                                                                           \
                if (two_tris_per_step(r)) { if (count_ref(r)) CL2_PERSIST(true, true, SRCT, SRC); else CL2_PERSIST(false, true, SRCT, SRC); } \
                else { if (count_ref(r)) CL2_PERSIST(true, false, SRCT, SRC); else CL2_PERSIST(false, false, SRCT, SRC); }       \
            } while (0)
            if (merged) CL2_PERSIST_SRC(DualPathRaySource, dual); else CL2_PERSIST_SRC(PathRaySource, src);
#undef CL2_PERSIST_SRC
#undef CL2_PERSIST
            r->launches_tp++;
            HIP_TRY(r, hipGetLastError());
            }
        }
        Timed t(r, split ? ST_BOUNCE : ST_TRAVERSE_PATHS, st);
        const float4* ext_hit = hits_ready ? r->d_hit_cam0 : r->d_hit;
#define CL2_TRACE(CAM, CNT, EXT)                                                                                          \
        hipLaunchKernelGGL((k_trace_subpath<CAM, CNT, EXT>), dim3(grid_for(B)), dim3(BLOCK), (EXT) ? 0 : bvh_lds_bytes(r), st, r->bvh, r->d_stats, \
                           first, end, q_in, c_in, q_out, c_out, B, pb, r->d_seeds, r->d_tri_shade, r->d_mats, r->n_mats,        \
                           r->d_block_stats, ext_hit)
        if (split) { if (which == CL2_CAMERA) CL2_TRACE(true, false, true); else CL2_TRACE(false, false, true); }
        else if (which == CL2_CAMERA) { if (count_ref(r)) CL2_TRACE(true, true, false); else CL2_TRACE(true, false, false); }
        else { if (count_ref(r)) CL2_TRACE(false, true, false); else CL2_TRACE(false, false, false); }
#undef CL2_TRACE
        if (!split) r->launches_tp++;
        HIP_TRY(r, hipGetLastError());
    }
    return CL2_OK;
}

// subpaths of every pixel: kinds = 1 light, 2 camera, 3 both (light first: one RNG stream per pixel)
int launch_subpaths(cl2_renderer* r, hipStream_t st, const PathBufs* set, int kinds) {
    if (!whole_subpaths(r)) {
        // both kinds in one call (cl2_run_samples): the camera subpaths' first rays ride in the light subpaths' level-0 launch
        const bool merge = kinds == 3;
        int rc = CL2_OK;
        if (kinds & 1) rc = launch_trace(r, CL2_LIGHT, st, set, merge ? 1 : 0);
        if (rc == CL2_OK && (kinds & 2)) rc = launch_trace(r, CL2_CAMERA, st, set, merge ? 2 : 0);
        return rc;
    }
    HIP_TRY(r, hipMemsetAsync(r->d_work, 0, WORK_STRIDE * sizeof(unsigned), st));
    Timed t(r, ST_TRAVERSE_PATHS, st);
    // lanes gathered / steps waited before a wave runs its bounce phase (cl2_set_subpath_gather)
    const int lanes = r->gather_lanes, wait = r->gather_wait;
    // the exact 4-wide walk inside the launch where the scene has it (same rule as the connection rays); round 4: also for trees
    // that stream from memory, with one triangle pair per pass there (as in launch_wide)
    const bool widew = wide_walk(r);
    WideView w = r->wide;
    w.stack_lds = 8; w.n_lds_nodes = 0;                        // 5 workgroups per CU here (registers): 24 KB each is free
    if (widew) {
        TRY(ensure_wide_overflow(r));
        w.n_lds_nodes = std::min(r->n_wide, 64);
        w.ovf_stride = std::max(r->wide_ovf_entries, 1);
        w.overflow = r->d_wide_ovf;                            // stage 0 region: the subpath stage
    }
    if ((r->debug_flags >> 14) & 1) w.tris36 = nullptr;         // bit 14: the 48-byte triangle records (bvh_wide.hpp: PACK off)
    const size_t lds = widew ? (size_t)w.stack_lds * BLOCK * 8 + (size_t)w.n_lds_nodes * 128 : bvh_lds_bytes(r);
    // register budget: 5 waves per SIMD (102 VGPRs: the bounce code; 6 waves cost 64 bytes of scratch per lane and measured
    // 13.0 -> 13.6 ms on the glass scene, 8 waves 24.5 ms); the grid holds as many workgroups as stay resident
    constexpr int WPS = 5;
    const int grid = std::max(1, persistent_grid_paths(r) * WPS / 8);
#define CL2_WHOLE(CNT, TWO, WIDE, ...)                                                                                    \
    hipLaunchKernelGGL((k_subpaths_persistent<CNT, TWO, WPS, WIDE, ##__VA_ARGS__>), dim3(grid), dim3(BLOCK), lds, st, r->bvh, w, \
                       r->B, r->d_work, set[CL2_LIGHT], set[CL2_CAMERA], r->d_seeds, r->d_tri_shade, r->d_mats, r->n_mats, \
                       r->d_stats, lanes, wait, kinds)
    if (widew) { if (two_tris_per_step(r)) CL2_WHOLE(false, true, true, WIDE_TRI_REPS); else CL2_WHOLE(false, true, true, 1); }
    else if (two_tris_per_step(r)) { if (count_ref(r)) CL2_WHOLE(true, true, false); else CL2_WHOLE(false, true, false); }
    else { if (count_ref(r)) CL2_WHOLE(true, false, false); else CL2_WHOLE(false, false, false); }
#undef CL2_WHOLE
    r->launches_tp++;
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

// connection set-up + connection rays; `cs` = which {chit, cmask} set
int launch_connect(cl2_renderer* r, hipStream_t st, const PathBufs* set, int cs) {
    const int B = r->B;
    const PathBufs& lp = set[CL2_LIGHT];
    const PathBufs& cp = set[CL2_CAMERA];
    HIP_TRY(r, hipMemsetAsync(r->d_qcount + 7, 0, sizeof(unsigned), st));
    {
        Timed t(r, ST_CONNECT_SETUP, st);
        // Large scenes: this launch (8,100 short workgroups at 7 waves per SIMD) runs beside the persistent subpath launches of
        // the next sample, whose chain is the critical path; left alone it floods the CUs and every level launch of that chain
        // starts late.  39 KB of unused dynamic LDS hold it to 4 workgroups per CU: glass 8.73 -> 8.46 ms per sample, blob 11.03
        // -> 10.85 (25 KB: 8.71 / 11.08; 31 KB: 8.44 / 10.98; 52 KB: 8.61 / 10.86).  The Cornell pipeline has no persistent
        // launches and wants the kernel at full speed.
        const size_t pad = split_conn(r) ? (size_t)39 * 1024 : 0;
        hipLaunchKernelGGL(k_connect_setup, dim3(grid_for(B)), dim3(BLOCK), pad, st, B, lp, cp,
                           r->d_mats, r->n_mats, r->cam, r->d_ctag, r->d_qcount + 7, r->d_cmask[cs]);
    }
    HIP_TRY(r, hipGetLastError());
    {
        Timed t(r, ST_TRAVERSE_CONN, st);
        if (split_conn(r)) {
            HIP_TRY(r, hipMemsetAsync(r->d_work + 7 * WORK_STRIDE, 0, WORK_STRIDE * sizeof(unsigned), st));
            ConnRaySource src{r->d_ctag, lp.P0, cp.P0, r->d_chit[cs],
                              V3{r->cam.focal_point[0], r->cam.focal_point[1], r->cam.focal_point[2]}, B};
            if (wide_walk(r)) {
                TRY(launch_wide(r, st, 1, r->d_qcount + 7, r->d_work + 7 * WORK_STRIDE, src, 1));
            } else {
#define CL2_PERSIST(CNT, TWO, SRCT, SRC)                                                                                  \
            hipLaunchKernelGGL((k_traverse_persistent<CNT, TWO, SRCT>), dim3(persistent_grid_conn(r)), dim3(BLOCK), bvh_lds_bytes(r), \
                               st, r->bvh, r->d_qcount + 7, r->d_work + 7 * WORK_STRIDE, SRC, r->d_stats, 1)
#define CL2_PERSIST_SRC(SRCT, SRC)                                                                                        \
            do {                                                                                                          \
                if (two_tris_per_step(r)) { if (count_ref(r)) CL2_PERSIST(true, true, SRCT, SRC); else CL2_PERSIST(false, true, SRCT, SRC); } \
                else { if (count_ref(r)) CL2_PERSIST(true, false, SRCT, SRC); else CL2_PERSIST(false, false, SRCT, SRC); }      \
            } while (0)
            CL2_PERSIST_SRC(ConnRaySource, src);
#undef CL2_PERSIST_SRC
#undef CL2_PERSIST
            }
        } else {
            // grid-stride over the (device-side) ray count; enough workgroups to fill 256 CUs several times over
            const int grid = std::min<size_t>(grid_for((size_t)B * 8), 256 * 32);
            if (count_ref(r))
                hipLaunchKernelGGL(k_traverse_conn<true>, dim3(grid), dim3(BLOCK), bvh_lds_bytes(r), st, r->bvh, B, r->d_qcount + 7, r->d_ctag,
                                   lp.P0, cp.P0, r->cam, r->d_chit[cs], r->d_stats);
            else
                hipLaunchKernelGGL(k_traverse_conn<false>, dim3(grid), dim3(BLOCK), bvh_lds_bytes(r), st, r->bvh, B, r->d_qcount + 7, r->d_ctag,
                                   lp.P0, cp.P0, r->cam, r->d_chit[cs], r->d_stats);
        }
        r->launches_tc++;
    }
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

// buffers of the reproducible light image: one record slot per (light vertex, entry), as the reference's five arrays
inline unsigned det_key_bits(const cl2_renderer* r) {          // bits of a target entry (< B) plus one for DET_NO_KEY's
    unsigned bits = 1;
    while (((size_t)1 << bits) < (size_t)r->B) bits++;
    return bits + 1;
}
int ensure_det_buffers(cl2_renderer* r) {
    // guarded by the LAST thing set up: a set whose allocation failed half way (CL2_E_NOMEM) is not mistaken for a complete one
    // by the next sample -- it is freed (dev_free nulls the pointers) and built again
    if (r->d_det_tmp) return CL2_OK;
    dev_free(r, r->d_det_keys); dev_free(r, r->d_det_keys_sorted); dev_free(r, r->d_det_slots); dev_free(r, r->d_det_slots_sorted); dev_free(r, r->d_det_vals);
    const size_t n = (size_t)MAX_VERTS * r->B;
    TRY(dev_alloc(r, &r->d_det_keys, n));
    TRY(dev_alloc(r, &r->d_det_keys_sorted, n));
    TRY(dev_alloc(r, &r->d_det_slots, n));
    TRY(dev_alloc(r, &r->d_det_slots_sorted, n));
    TRY(dev_alloc(r, &r->d_det_vals, n));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(BLOCK), 0, r->stream, r->d_det_slots, n);
    HIP_TRY(r, hipGetLastError());
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    size_t bytes = 0;
    HIP_TRY(r, det_sort_pairs(nullptr, bytes, r->d_det_keys, r->d_det_keys_sorted, r->d_det_slots, r->d_det_slots_sorted, n, det_key_bits(r), r->stream));
    unsigned char* tmp = nullptr;
    TRY(dev_alloc(r, &tmp, bytes));
    r->d_det_tmp = tmp; r->det_tmp_bytes = bytes;
    return CL2_OK;
}

int launch_resolve(cl2_renderer* r, hipStream_t st, const PathBufs* set, int cs) {
    const int B = r->B;
    const PathBufs& lp = set[CL2_LIGHT];
    const PathBufs& cp = set[CL2_CAMERA];
    {
        Timed t(r, ST_CONNECT_RESOLVE, st);
#define CL2_RESOLVE(W, ML, DET)                                                                                                  \
        hipLaunchKernelGGL((k_connect_resolve<W, ML, DET>), dim3(grid_for(B)), dim3(BLOCK), 0, st, B, lp, cp, r->d_mats, r->n_mats, r->d_tri_shade, \
                           r->cam_tris, r->cam, r->d_cmask[cs], r->d_chit[cs], r->d_agg, r->d_light_image, r->d_uni, r->d_stats, r->debug_flags, \
                           r->d_det_keys, r->d_det_vals)
        // 3 waves per SIMD: what 165 VGPRs and 52 KB of LDS tables per workgroup allow (2 / 4 measured slower: DESIGN 6.1).  Debug
        // bits 4-6 = 7: one wave per camera vertex (tests/connect_resolve_wide.hpp: same results bit for bit, measured slower: 1.14 vs
        // 0.93 ms; a second implementation kept as a cross-check, built only with -DCL2_TEST_VARIANT = libclive2_amd_test.so)
        const int occ = (r->debug_flags >> 4) & 7;
        if (occ != 0 && occ != 7) return fail(r, CL2_E_INVALID, "debug bits 4-6 must be 0 or 7");
        if (occ == 7 && r->streams != 1) return fail(r, CL2_E_INVALID, "the cross-check resolve kernel handles one sample stream");
        // (it splats with atomics and writes no records: the sort + gather below would read buffers it never filled)
        if (occ == 7 && r->reproducible) return fail(r, CL2_E_INVALID, "the cross-check resolve kernel has no reproducible form: clear debug bits 4-6 or cl2_set_reproducible(0)");
#ifdef CL2_TEST_VARIANT
        if (occ == 7)
            hipLaunchKernelGGL(k_connect_resolve_wide, dim3((B + RW_PIX - 1) / RW_PIX), dim3(RW_BLOCK), 0, st, B, lp, cp, r->d_mats,
                               r->d_tri_shade, r->cam, r->d_cmask[cs], r->d_chit[cs], r->d_agg, r->d_light_image, r->d_uni,
                               r->d_stats, r->debug_flags);
        else
#else
        if (occ == 7) return fail(r, CL2_E_INVALID, "the one-wave-per-camera-vertex resolve kernel is only built into the test variant of the library");
#endif
        // the material table goes to LDS when it fits LDS_MAT_CAP entries (the reference ships 8): connect_resolve.hpp
        if (r->reproducible) {
            // records instead of atomics: every slot starts empty (DET_NO_KEY = all ones)
            TRY(ensure_det_buffers(r));
            HIP_TRY(r, hipMemsetAsync(r->d_det_keys, 0xFF, (size_t)MAX_VERTS * B * sizeof(unsigned), st));
            if (r->n_mats <= LDS_MAT_CAP) CL2_RESOLVE(2, true, true); else CL2_RESOLVE(2, false, true);      // the record stores need registers the 168 of three waves do not leave
        } else {
            // (more than LDS_MAT_CAP materials: the table stays in global memory and its addresses take the registers that three
            // waves per SIMD do not leave -- two waves, no scratch)
            if (r->n_mats <= LDS_MAT_CAP) CL2_RESOLVE(3, true, false); else CL2_RESOLVE(2, false, false);
        }
#undef CL2_RESOLVE
    }
    HIP_TRY(r, hipGetLastError());
    if (r->reproducible) {
        // a stable sort of {target entry, slot} on the target's bits: a target's run is then ordered by slot = (s, source pixel);
        // one thread per target sums its run front to back
        Timed t(r, ST_CONNECT_RESOLVE, st);
        const size_t n = (size_t)MAX_VERTS * B;
        size_t bytes = r->det_tmp_bytes;
        HIP_TRY(r, det_sort_pairs(r->d_det_tmp, bytes, r->d_det_keys, r->d_det_keys_sorted, r->d_det_slots, r->d_det_slots_sorted, n,
                                  det_key_bits(r), st));
        hipLaunchKernelGGL(k_det_gather, dim3(grid_for(n)), dim3(BLOCK), 0, st, r->d_det_keys_sorted, r->d_det_slots_sorted, n,
                           r->d_det_vals, r->d_light_image);
        HIP_TRY(r, hipGetLastError());
    }
    return CL2_OK;
}

int launch_finalize(cl2_renderer* r, hipStream_t st) {
    Timed t(r, ST_FINALIZE, st);
    hipLaunchKernelGGL(k_finalize, dim3(grid_for(r->B)), dim3(BLOCK), 0, st, r->B, r->W, r->H, r->d_agg, r->d_finalized,
                       r->d_sample_w);
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

int launch_accumulate(cl2_renderer* r, hipStream_t st) {
    Timed t(r, ST_ACCUMULATE, st);
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(r->FB)), dim3(BLOCK), 0, st, r->FB, r->streams, r->d_finalized, r->d_sample_w,
                       r->d_light_image, r->d_uni, r->d_acc);
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

int launch_finalize_accumulate(cl2_renderer* r, hipStream_t st) {
    Timed t(r, ST_FINALIZE, st);
    hipLaunchKernelGGL(k_finalize_accumulate, dim3(grid_for(r->FB)), dim3(BLOCK), 0, st, r->B, r->W, r->H, r->d_agg, r->d_light_image,
                       r->d_uni, r->d_acc);
    HIP_TRY(r, hipGetLastError());
    return CL2_OK;
}

int need_scene(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    r->pipe_active = false;
    if (!r->scene_ok) return fail(r, CL2_E_STATE, "no scene uploaded (call cl2_upload_scene first)");
    HIP_TRY(r, hipSetDevice(r->device));
    return CL2_OK;
}

// Everything that has one entry per (sample stream, pixel): allocated for r->B = r->streams x r->FB entries and initialised
// as cl2_create always did (seeds 1, zero lengths / images / masks).  cl2_set_sample_streams frees and re-allocates it.
void free_pixel_state(cl2_renderer* r) {
    dev_free(r, r->d_seeds);
    for (int q = 0; q < 3; q++)
        for (int k = 0; k < 2; k++) {
            PathBufs& pb = r->sets[q][k];
            dev_free(r, pb.P0); dev_free(r, pb.P1); dev_free(r, pb.P2); dev_free(r, pb.P3);
            dev_free(r, pb.tri); dev_free(r, pb.len); dev_free(r, pb.carry);
        }
    dev_free(r, r->d_hit); dev_free(r, r->d_hit_cam0); dev_free(r, r->d_queue); dev_free(r, r->d_ctag);
    dev_free(r, r->d_det_keys); dev_free(r, r->d_det_keys_sorted); dev_free(r, r->d_det_slots); dev_free(r, r->d_det_slots_sorted);
    dev_free(r, r->d_det_vals); dev_free(r, r->d_det_tmp);
    r->det_tmp_bytes = 0;
    for (int q = 0; q < 2; q++) { dev_free(r, r->d_chit[q]); dev_free(r, r->d_cmask[q]); }
    dev_free(r, r->d_agg); dev_free(r, r->d_light_image); dev_free(r, r->d_finalized); dev_free(r, r->d_uni);
    dev_free(r, r->d_sample_w); dev_free(r, r->d_block_stats);
}

int alloc_pixel_state(cl2_renderer* r) {
    const size_t B = (size_t)r->B;
    int rc = CL2_OK;
#define A(ptr, count) if (rc == CL2_OK) rc = dev_alloc(r, &(ptr), (count))
    A(r->d_seeds, B);
    for (int q = 0; q < 3; q++) {
        for (int k = 0; k < 2; k++) {
            PathBufs& pb = r->sets[q][k];
            A(pb.P0, MAX_VERTS * B); A(pb.P1, MAX_VERTS * B); A(pb.P2, MAX_VERTS * B); A(pb.P3, MAX_VERTS * B);
            A(pb.tri, MAX_VERTS * B); A(pb.len, B); A(pb.carry, B);
        }
    }
    A(r->d_hit, B);
    A(r->d_hit_cam0, B);
    A(r->d_queue, MAX_VERTS * B);
    A(r->d_ctag, (size_t)CONN_SLOTS * B);
    for (int q = 0; q < 2; q++) { A(r->d_chit[q], (size_t)CONN_SLOTS * B); A(r->d_cmask[q], B); }
    A(r->d_agg, (size_t)AGG_ROWS * B);
    A(r->d_light_image, B); A(r->d_finalized, B); A(r->d_uni, B);
    A(r->d_sample_w, B);
    A(r->d_block_stats, (size_t)grid_for(B) * 4);
#undef A
    if (rc != CL2_OK) return rc;
    unsigned qc[9] = {(unsigned)r->B, 0, 0, 0, 0, 0, 0, 0, 2u * (unsigned)r->B};      // [8]: ray count of the merged level-0 launch
    bool ok = hipMemcpy(r->d_qcount, qc, sizeof qc, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(r->d_light_image, 0, B * sizeof(float4)) == hipSuccess;
    ok = ok && hipMemset(r->d_finalized, 0, B * sizeof(float4)) == hipSuccess;
    ok = ok && hipMemset(r->d_uni, 0, B * sizeof(float4)) == hipSuccess;
    ok = ok && hipMemset(r->d_sample_w, 0, B * sizeof(float)) == hipSuccess;
    ok = ok && hipMemset(r->d_agg, 0, (size_t)AGG_ROWS * B * sizeof(float)) == hipSuccess;
    ok = ok && hipMemset(r->d_block_stats, 0, (size_t)grid_for(B) * 4 * sizeof(unsigned long long)) == hipSuccess;
    for (int q = 0; q < 2 && ok; q++) ok = hipMemset(r->d_cmask[q], 0, B * sizeof(unsigned long long)) == hipSuccess;
    for (int q = 0; q < 3 && ok; q++)
        for (int k = 0; k < 2 && ok; k++) ok = hipMemset(r->sets[q][k].len, 0, B * sizeof(int)) == hipSuccess;
    // default seeds: 1 everywhere (xorshift's only forbidden state is 0); callers set real seeds
    if (ok) {
        std::vector<uint32_t> ones(2 * B, 1u);
        ok = hipMemcpy(r->d_seeds, ones.data(), 2 * B * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) return fail(r, CL2_E_HIP, "device buffer initialisation failed");
    r->cur = 0;
    return CL2_OK;
}

#define STAGE_PROLOGUE(r)                 \
    do {                                  \
        int rc_ = need_scene(r);          \
        if (rc_ != CL2_OK) return rc_;    \
    } while (0)

}  // namespace

extern "C" {

int cl2_abi_version(void) { return 5; }

int cl2_build_bvh(const double* tri_min, const double* tri_max, int64_t n_triangles, int max_members, int max_depth,
                  void* out_boxes, int64_t box_capacity, int64_t* n_boxes_out, int64_t* out_perm) {
    if (!tri_min || !tri_max || !out_boxes || !n_boxes_out || !out_perm || n_triangles < 1 || max_members < 1 ||
        n_triangles > (int64_t)1 << 30) {
        g_create_error = "cl2_build_bvh: bad argument";
        return CL2_E_INVALID;
    }
    BvhBuildResult res;
    try {
        res = build_bvh_sah(tri_min, tri_max, n_triangles, max_members, max_depth);
    } catch (const std::exception& ex) {
        g_create_error = std::string("cl2_build_bvh: ") + ex.what();
        return CL2_E_NOMEM;
    }
    if ((int64_t)res.boxes.size() > box_capacity) { g_create_error = "cl2_build_bvh: box_capacity too small"; return CL2_E_INVALID; }
    static_assert(sizeof(HostBox) == 48, "Box record");
    std::memcpy(out_boxes, res.boxes.data(), res.boxes.size() * sizeof(HostBox));
    std::memcpy(out_perm, res.perm.data(), res.perm.size() * sizeof(int64_t));
    *n_boxes_out = (int64_t)res.boxes.size();
    return CL2_OK;
}

void cl2_set_create_error(const char* msg) { g_create_error = msg ? msg : ""; }   // for the entry points of other translation units

const char* cl2_last_error(const cl2_renderer* r) { return r ? r->err.c_str() : g_create_error.c_str(); }

int cl2_create(int device_ordinal, int pixel_width, int pixel_height, cl2_renderer** out) {
    if (!out) { g_create_error = "out is NULL"; return CL2_E_INVALID; }
    *out = nullptr;
    if (pixel_width < 1 || pixel_height < 1 || (long long)pixel_width * pixel_height >= (1ll << TAG_PID_BITS)) {     // also checked per stream count
        g_create_error = "frame size out of range (need 1 <= W*H < 2^26)";
        return CL2_E_INVALID;
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1) {
        g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return CL2_E_HIP;
    }
    if (device_ordinal < 0 || device_ordinal >= n_dev) { g_create_error = "device ordinal out of range"; return CL2_E_INVALID; }
    cl2_renderer* r = new cl2_renderer();
    r->device = device_ordinal;
    r->W = pixel_width; r->H = pixel_height; r->FB = pixel_width * pixel_height;
    r->streams = 1; r->B = r->FB;
    auto bail = [&](int code) { g_create_error = r->err; cl2_destroy(r); return code; };
    if (hipSetDevice(device_ordinal) != hipSuccess) { r->err = "hipSetDevice failed"; return bail(CL2_E_HIP); }
    if (hipStreamCreate(&r->stream) != hipSuccess || hipStreamCreate(&r->stream_conn) != hipSuccess ||
        hipStreamCreate(&r->stream_res) != hipSuccess) {
        r->err = "hipStreamCreate failed"; return bail(CL2_E_HIP);
    }
    {
        bool ok_ev = true;
        for (auto& e : r->ev_paths) ok_ev = ok_ev && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto& e : r->ev_conn) ok_ev = ok_ev && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto& e : r->ev_res) ok_ev = ok_ev && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto& e : r->ev_tune) ok_ev = ok_ev && hipEventCreate(&e) == hipSuccess;
        if (!ok_ev) { r->err = "hipEventCreate failed"; return bail(CL2_E_HIP); }
    }
    int rc = CL2_OK;
    if (rc == CL2_OK) rc = dev_alloc(r, &r->d_qcount, (size_t)9);
    if (rc == CL2_OK) rc = dev_alloc(r, &r->d_work, (size_t)8 * WORK_STRIDE);
    if (rc == CL2_OK) rc = dev_alloc(r, &r->d_acc, 8 * (size_t)r->FB);
    if (rc == CL2_OK) rc = dev_alloc(r, &r->d_stats, (size_t)1);
    if (rc != CL2_OK) return bail(rc);
    bool ok = hipMemset(r->d_acc, 0, 8 * (size_t)r->FB * sizeof(float)) == hipSuccess;
    ok = ok && hipMemset(r->d_stats, 0, sizeof(Stats)) == hipSuccess;
    if (!ok) { r->err = "device buffer initialisation failed"; return bail(CL2_E_HIP); }
    rc = alloc_pixel_state(r);
    if (rc != CL2_OK) return bail(rc);
    *out = r;
    return CL2_OK;
}

void cl2_destroy(cl2_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    // A handle whose collective failed or timed out was torn down with ncclCommAbort; whether that retires the collective's
    // kernel on r->stream is RCCL's business (unverified on hardware: RCCL has never run here with more than one rank), and
    // an unbounded hipStreamSynchronize on that stream would turn "exit non-zero" into "hang in destroy" (ADVICE r3).  Such a
    // handle polls its streams for a few seconds and then LEAKS its device resources (hipFree would wait as well): the
    // process is on its way out with an error.
    const bool poisoned = r->comm_poisoned;
    if (r->comm) (void)cl2_comm_destroy(r);
    auto settle = [&](hipStream_t s) -> bool {
        if (!s) return true;
        if (!poisoned) { (void)hipStreamSynchronize(s); return true; }
        const auto until = std::chrono::steady_clock::now() + std::chrono::seconds(5);
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return true;          // drained, or in an error state: nothing left to wait for
            if (std::chrono::steady_clock::now() > until) return false;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    };
    const bool s0 = settle(r->stream), s1 = settle(r->stream_conn), s2 = settle(r->stream_res);   // every stream gets its poll
    if (!(s0 && s1 && s2)) {
        std::fprintf(stderr, "clive2_amd: cl2_destroy: a stream did not drain after the communicator was aborted; leaking the handle\n");
        return;
    }
    for (auto e : r->ev_paths) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_conn) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_res) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_tune) if (e) (void)hipEventDestroy(e);
    for (auto& s : r->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : r->event_pool) (void)hipEventDestroy(e);
    for (void* p : r->allocs) (void)hipFree(p);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    if (r->stream_conn) (void)hipStreamDestroy(r->stream_conn);
    if (r->stream_res) (void)hipStreamDestroy(r->stream_res);
    delete r;
}

int cl2_upload_scene(cl2_renderer* r, const void* boxes_v, int n_boxes, const void* tris_v, int n_tris, const void* mats_v,
                     int n_mats, const void* camera_v, const void* light_tris_v, const float* light_areas,
                     const int32_t* light_tri_index, int light_count) {
    if (!r) return CL2_E_INVALID;
    if (!boxes_v || !tris_v || !mats_v || !camera_v || !light_tris_v || !light_areas || !light_tri_index)
        return fail(r, CL2_E_INVALID, "NULL scene array");
    if (n_boxes < 1 || n_tris < 1 || light_count < 1) return fail(r, CL2_E_INVALID, "scene needs >=1 box, triangle and light");
    if (n_tris >= (1 << 27)) return fail(r, CL2_E_INVALID, "at most 2^27 triangles (leaf records pack begin<<4 | count-1)");
    // material 7 is hard-wired into the camera vertices (trace.metal:611, :1053); at most 256 fit the packed meta word
    if (n_mats < 8 || n_mats > 256) return fail(r, CL2_E_INVALID, "material table must have 8..256 entries");
    const BoxRec* boxes = static_cast<const BoxRec*>(boxes_v);
    const TriRec* tris = static_cast<const TriRec*>(tris_v);
    const MatRec* mats = static_cast<const MatRec*>(mats_v);
    const TriRec* ltris = static_cast<const TriRec*>(light_tris_v);
    CameraRec cam;
    std::memcpy(&cam, camera_v, sizeof cam);
    if (cam.pixel_width != r->W || cam.pixel_height != r->H)
        return fail(r, CL2_E_INVALID, "camera resolution differs from the renderer's");

    // ---- validate the tree: every index in range, children after their parent (breadth-first
    // numbering, src/bvh.py:345-351) and every box reached exactly once -- which also guarantees
    // that the stackless walk terminates ----
    std::vector<char> reached(n_boxes, 0);
    reached[0] = 1;
    for (int i = 0; i < n_boxes; i++) {
        const BoxRec& b = boxes[i];
        if (!reached[i]) return fail(r, CL2_E_INVALID, "box " + std::to_string(i) + " is not reachable from the root");
        if (b.right == 0) {
            if (b.left <= i || b.left + 1 >= n_boxes) return fail(r, CL2_E_INVALID, "inner box child index out of order/range");
            if (reached[b.left] || reached[b.left + 1]) return fail(r, CL2_E_INVALID, "box has two parents");
            reached[b.left] = reached[b.left + 1] = 1;
        } else {
            if (b.left < 0 || b.right > n_tris || b.left >= b.right) return fail(r, CL2_E_INVALID, "leaf triangle range out of range");
        }
    }
    // ---- visit order (node, right subtree, left subtree = the reference's pop order, trace.metal:150-160)
    // and subtree sizes in records; leaves with more than LEAF_PACK_MAX triangles take extra records ----
    auto leaf_records = [&](const BoxRec& b) { return (b.right - b.left + LEAF_PACK_MAX - 1) / LEAF_PACK_MAX; };
    std::vector<int> subtree(n_boxes, 0);
    for (int i = n_boxes - 1; i >= 0; i--) {            // children have larger indices than their parent
        const BoxRec& b = boxes[i];
        subtree[i] = b.right == 0 ? 1 + subtree[b.left] + subtree[b.left + 1] : leaf_records(b);
    }
    const int n_records = subtree[0];
    std::vector<int> rec_index(n_boxes, -1);
    // `pending[i]`: entries on the reference's stack underneath box i when it is popped.  The reference's loop
    // runs `while (stack_ptr > 0 && stack_ptr < 64)` (trace.metal:149): a walk that enters an inner box with 62
    // entries pending pushes to 64 and ENDS there, whatever is still unvisited (quirk Q18).  The stackless walk has
    // no such limit, so a tree that could reach it is refused instead of being rendered differently; the
    // reference's builder stops splitting at 32 pending boxes (bvh.py:294, Q13), far below.
    std::vector<int> pending(n_boxes, 0);
    rec_index[0] = 0;
    for (int i = 0; i < n_boxes; i++) {                  // parents before children: their record index is known
        const BoxRec& b = boxes[i];
        if (b.right == 0) {
            if (pending[i] + 2 >= 64)
                return fail(r, CL2_E_INVALID, "tree too deep: the reference's 64-entry traversal stack would overflow at box " + std::to_string(i));
            rec_index[b.left + 1] = rec_index[i] + 1;                            // right child: adjacent
            rec_index[b.left] = rec_index[i] + 1 + subtree[b.left + 1];          // left child: after the right subtree
            pending[b.left + 1] = pending[i] + 1;                                // popped first, its sibling waits below it
            pending[b.left] = pending[i];
        }
    }
    // ---- record numbering.  Small trees: plain visit order.  Trees larger than the LDS window: the
    // boxes of the TOP levels (the reference array is breadth-first, so a prefix of it) are numbered
    // first, [0, n_top), so that the window staged in LDS holds the records every ray visits; the rest
    // keep their visit order behind them.  Links are explicit (skip, and the right child in `info`),
    // so the numbering has no influence on the walk. ----
    int n_top = 0;
    if (n_records > LDS_NODE_CAP) {
        while (n_top < n_boxes && n_top < LDS_NODE_CAP && (boxes[n_top].right == 0 || leaf_records(boxes[n_top]) == 1)) n_top++;
    }
    std::vector<int> renum((size_t)n_records + 1);
    {
        std::vector<char> is_top((size_t)n_records, 0);
        for (int i = 0; i < n_top; i++) is_top[rec_index[i]] = 1;
        int tops_before = 0;
        for (int k = 0; k < n_records; k++) {
            if (is_top[k]) { tops_before++; continue; }
            renum[k] = n_top + k - tops_before;
        }
        for (int i = 0; i < n_top; i++) renum[rec_index[i]] = i;
        renum[n_records] = n_records;
    }
    for (int t = 0; t < n_tris; t++) {
        if (tris[t].material < 0 || tris[t].material >= n_mats) return fail(r, CL2_E_INVALID, "triangle material index out of range");
    }
    for (int l = 0; l < light_count; l++) {
        if (light_tri_index[l] < 0 || light_tri_index[l] >= n_tris) return fail(r, CL2_E_INVALID, "light triangle index out of range");
        if (ltris[l].material < 0 || ltris[l].material >= n_mats) return fail(r, CL2_E_INVALID, "light material index out of range");
    }

    // ---- 4-wide collapse for the exact wide walk (bvh_wide.hpp).  Conditions: the root is an inner box, every box
    // nests its children exactly (what the exactness argument rests on; true for trees that np_flatten_bvh or either
    // native builder made, not guaranteed for hand-made Box[] arrays) and no leaf exceeds one record. ----
    std::vector<float4> h_wide;
    int n_wide = 0;
    bool nests = false;
    {
        bool ok = n_boxes >= 3 && boxes[0].right == 0;
        auto inside = [&](const BoxRec& c, const BoxRec& p) {
            for (int k = 0; k < 3; k++) if (!(c.min[k] >= p.min[k] && c.max[k] <= p.max[k])) return false;
            return true;
        };
        for (int i = 0; i < n_boxes && ok; i++) {
            const BoxRec& b = boxes[i];
            if (b.right == 0) ok = inside(boxes[b.left], b) && inside(boxes[b.left + 1], b);
            else ok = (b.right - b.left) <= LEAF_PACK_MAX;
        }
        nests = ok;
        if (ok) {
            // slots of reference box x in its visit order: child left+1 first, each inner child replaced by its children
            auto slots_of = [&](int x, int* out) {
                int n = 0;
                for (int c : {boxes[x].left + 1, boxes[x].left}) {
                    if (boxes[c].right != 0) out[n++] = c;
                    else { out[n++] = boxes[c].left + 1; out[n++] = boxes[c].left; }
                }
                return n;
            };
            std::vector<int> wide_of(n_boxes, -1), order;
            order.push_back(0); wide_of[0] = 0;
            for (size_t h = 0; h < order.size(); h++) {
                int sl[4];
                const int n = slots_of(order[h], sl);
                for (int k = 0; k < n; k++)
                    if (boxes[sl[k]].right == 0) { wide_of[sl[k]] = (int)order.size(); order.push_back(sl[k]); }
            }
            n_wide = (int)order.size();
            h_wide.assign((size_t)8 * n_wide, make_float4(0, 0, 0, 0));
            for (int wn = 0; wn < n_wide; wn++) {
                int sl[4];
                const int n = slots_of(order[wn], sl);
                float v[6][4];
                int ref[4];
                for (int k = 0; k < 4; k++) {
                    ref[k] = WIDE_EMPTY;
                    // an empty slot holds a box at +inf: for a ray with finite 1/d its slab test gives tmin = +inf or tmax = -inf,
                    // so `tmin <= tmax && tmin < best_t` fails without the walk looking at the reference (bvh_wide.hpp)
                    for (int c = 0; c < 6; c++) v[c][k] = std::numeric_limits<float>::infinity();
                    if (k >= n) continue;
                    const BoxRec& b = boxes[sl[k]];
                    for (int c = 0; c < 3; c++) { v[c][k] = b.min[c]; v[3 + c][k] = b.max[c]; }
                    ref[k] = b.right == 0 ? wide_of[sl[k]] : ~((b.left << 4) | (b.right - b.left - 1));
                }
                for (int c = 0; c < 6; c++) h_wide[(size_t)8 * wn + c] = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
                h_wide[(size_t)8 * wn + 6] = make_float4(as_f_int(ref[0]), as_f_int(ref[1]), as_f_int(ref[2]), as_f_int(ref[3]));
            }
        }
    }

    // ---- repack ----
    std::vector<float4> h_nodes(2 * (size_t)n_records), h_tris(3 * (size_t)n_tris), h_shade(4 * (size_t)n_tris),
        h_ltris(5 * (size_t)light_count);
    auto as_f = [](int32_t i) { float f; std::memcpy(&f, &i, 4); return f; };
    const float inf = std::numeric_limits<float>::infinity();
    for (int i = 0; i < n_boxes; i++) {
        const BoxRec& b = boxes[i];
        const int k = rec_index[i], skip = renum[k + subtree[i]];
        if (b.right == 0) {
            // inner: info = ~(record of the right child, visited first), negative
            h_nodes[2 * (size_t)renum[k]] = make_float4(b.min[0], b.min[1], b.min[2], as_f(skip));
            h_nodes[2 * (size_t)renum[k] + 1] = make_float4(b.max[0], b.max[1], b.max[2], as_f(~renum[k + 1]));
        } else {
            int begin = b.left;
            for (int part = 0; begin < b.right; part++, begin += LEAF_PACK_MAX) {
                const int count = std::min(LEAF_PACK_MAX, b.right - begin);
                const int info = (begin << 4) | (count - 1);
                // follow-up records of an oversized leaf: an unbounded box, so they are always entered
                const int nxt = renum[k + part + 1];
                const float4 lo = part == 0 ? make_float4(b.min[0], b.min[1], b.min[2], as_f(nxt))
                                            : make_float4(-inf, -inf, -inf, as_f(nxt));
                const float4 hi = part == 0 ? make_float4(b.max[0], b.max[1], b.max[2], as_f(info))
                                            : make_float4(inf, inf, inf, as_f(info));
                h_nodes[2 * (size_t)renum[k + part]] = lo;
                h_nodes[2 * (size_t)renum[k + part] + 1] = hi;
            }
        }
    }
    // ---- pruned table for the LDS-resident walk.  For a ray with finite 1/d an inner box's test can only prune (same
    // argument as the wide walk: a child that passes its test implies its parent passed), so an inner record may be
    // dropped and its children visited unconditionally without changing any hit.  A record is dropped when the test is
    // expected to cost more than it saves: (1 - area / area of the nearest tested ancestor) x cost of the subtree < 1
    // box test, the surface-area estimate of the chance that a ray that reached the ancestor misses this box.  The
    // Cornell box loses its root and its one other inner box (both span the room): 3 box tests per ray instead of 5.
    // Rays with a non-finite 1/d, and the counting mode, walk the full table. ----
    std::vector<float4> h_fast;
    int n_fast = 0;
    if (nests && n_top == 0 && n_records <= LDS_NODE_CAP && n_tris <= LDS_TRI_CAP) {
        auto area = [&](const BoxRec& b) {
            const double x = (double)b.max[0] - b.min[0], y = (double)b.max[1] - b.min[1], z = (double)b.max[2] - b.min[2];
            return 2.0 * (x * y + y * z + z * x);
        };
        std::vector<double> cost(n_boxes, 0.0), anc_area(n_boxes, 0.0);
        for (int i = n_boxes - 1; i >= 0; i--) {
            const BoxRec& b = boxes[i];
            cost[i] = b.right == 0 ? 1.0 + cost[b.left] + cost[b.left + 1] : 1.0 + 2.5 * (b.right - b.left);
        }
        std::vector<char> dropped((size_t)n_records + 1, 0);
        anc_area[0] = area(boxes[0]);
        for (int i = 0; i < n_boxes; i++) {
            const BoxRec& b = boxes[i];
            if (b.right != 0) continue;
            const double a = area(b);
            const double p_miss = anc_area[i] > 0.0 ? std::max(0.0, 1.0 - a / anc_area[i]) : 0.0;
            const bool drop = p_miss * (cost[i] - 1.0) < 1.0;
            dropped[rec_index[i]] = drop ? 1 : 0;
            anc_area[b.left] = anc_area[b.left + 1] = drop ? anc_area[i] : a;
        }
        std::vector<int> fast_index((size_t)n_records + 1);
        for (int k = 0; k <= n_records; k++) { fast_index[k] = n_fast; if (k < n_records && !dropped[k]) n_fast++; }
        // LDS budget: the pruned table sits beside the full one (rays with a non-finite 1/d need that) in every workgroup
        // that stages the tree.  Three such workgroups per CU (160 KB) is what the subpath kernel runs at with its 9.7 KB
        // of static shading tables; a scene near the 512-record / 512-triangle caps would lose a workgroup per CU to the
        // extra table (and a 64-KB-per-workgroup part would refuse the launch), so there it is not built.
        const size_t lds_with_fast = ((size_t)2 * n_records + (size_t)3 * n_tris + (size_t)2 * n_fast) * sizeof(float4) + sizeof(ShadeLds);
        if (n_fast < n_records && lds_with_fast > (size_t)160 * 1024 / 3) n_fast = n_records;       // -> no table
        if (n_fast < n_records) {
            h_fast.resize(2 * (size_t)n_fast);
            for (int i = 0; i < n_boxes; i++) {
                const int k = rec_index[i];
                if (dropped[k]) continue;
                float4 lo = h_nodes[2 * (size_t)k], hi = h_nodes[2 * (size_t)k + 1];
                lo.w = as_f(fast_index[k + subtree[i]]);
                if (boxes[i].right == 0) hi.w = as_f(~fast_index[k + 1]);
                h_fast[2 * (size_t)fast_index[k]] = lo;
                h_fast[2 * (size_t)fast_index[k] + 1] = hi;
            }
        } else {
            n_fast = 0;
        }
    }

    CamTris cam_tris{0, {0, 0, 0, 0}};
    for (int t = 0; t < n_tris; t++) {
        if (!tris[t].is_camera || cam_tris.n < 0) continue;
        if (cam_tris.n == CAM_TRI_ARGS) cam_tris.n = -1;
        else cam_tris.idx[cam_tris.n++] = t;
    }
    for (int t = 0; t < n_tris; t++) {
        const TriRec& T = tris[t];
        // edge vectors: the same binary32 subtractions ray_triangle_intersect performs (trace.metal:118-119)
        h_tris[3 * t] = make_float4(T.v0[0], T.v0[1], T.v0[2], 0.0f);
        h_tris[3 * t + 1] = make_float4(T.v1[0] - T.v0[0], T.v1[1] - T.v0[1], T.v1[2] - T.v0[2], 0.0f);
        h_tris[3 * t + 2] = make_float4(T.v2[0] - T.v0[0], T.v2[1] - T.v0[1], T.v2[2] - T.v0[2], 0.0f);
        h_shade[4 * t] = make_float4(T.n0[0], T.n0[1], T.n0[2], as_f(T.material));
        h_shade[4 * t + 1] = make_float4(T.n1[0], T.n1[1], T.n1[2], as_f(T.is_light ? 1 : 0));
        h_shade[4 * t + 2] = make_float4(T.n2[0], T.n2[1], T.n2[2], as_f(T.is_camera ? 1 : 0));
        h_shade[4 * t + 3] = make_float4(T.normal[0], T.normal[1], T.normal[2], 0.0f);
    }
    for (int l = 0; l < light_count; l++) {
        const TriRec& T = ltris[l];
        h_ltris[5 * l] = make_float4(T.v0[0], T.v0[1], T.v0[2], 0.0f);
        h_ltris[5 * l + 1] = make_float4(T.v1[0], T.v1[1], T.v1[2], 0.0f);
        h_ltris[5 * l + 2] = make_float4(T.v2[0], T.v2[1], T.v2[2], 0.0f);
        h_ltris[5 * l + 3] = make_float4(T.normal[0], T.normal[1], T.normal[2], 0.0f);
        h_ltris[5 * l + 4] = make_float4(as_f(T.material), 0.0f, 0.0f, 0.0f);
    }
    std::vector<MaterialDev> h_mats(n_mats);
    for (int m = 0; m < n_mats; m++) {
        h_mats[m].color_type = make_float4(mats[m].color[0], mats[m].color[1], mats[m].color[2], as_f(mats[m].type));
        h_mats[m].emission_alpha = make_float4(mats[m].emission[0], mats[m].emission[1], mats[m].emission[2], mats[m].alpha);
        h_mats[m].ior = mats[m].ior;
        h_mats[m].pad[0] = h_mats[m].pad[1] = h_mats[m].pad[2] = 0.0f;
    }

    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    r->scene_ok = false;
    dev_free(r, r->d_nodes); dev_free(r, r->d_tris); dev_free(r, r->d_tri_shade);
    dev_free(r, r->d_mats); dev_free(r, r->d_light_tris); dev_free(r, r->d_light_areas); dev_free(r, r->d_light_tri_index);
    TRY(dev_alloc(r, &r->d_nodes, h_nodes.size()));
    TRY(dev_alloc(r, &r->d_tris, h_tris.size()));
    TRY(dev_alloc(r, &r->d_tri_shade, h_shade.size()));
    TRY(dev_alloc(r, &r->d_mats, h_mats.size()));
    TRY(dev_alloc(r, &r->d_light_tris, h_ltris.size()));
    TRY(dev_alloc(r, &r->d_light_areas, (size_t)light_count));
    TRY(dev_alloc(r, &r->d_light_tri_index, (size_t)light_count));
    HIP_TRY(r, hipMemcpy(r->d_nodes, h_nodes.data(), h_nodes.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_tris, h_tris.data(), h_tris.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_tri_shade, h_shade.data(), h_shade.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_mats, h_mats.data(), h_mats.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_light_tris, h_ltris.data(), h_ltris.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_light_areas, light_areas, (size_t)light_count * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(r, hipMemcpy(r->d_light_tri_index, light_tri_index, (size_t)light_count * sizeof(int), hipMemcpyHostToDevice));

    dev_free(r, r->d_wide);
    dev_free(r, r->d_tris36);
    r->wide.tris36 = nullptr;
    dev_free(r, r->d_tri_rank);
    r->wide.tri_rank = nullptr;
    r->n_wide = 0;
    if (n_wide > 0) {
        // the wide walk reads the triangle records without their three padding words (36 bytes each; bvh_wide.hpp, PACK)
        {
            std::vector<float> h36((size_t)9 * n_tris + 9, 0.0f);          // + one record of padding: the pair load of the last triangle reads it
            for (int t = 0; t < n_tris; t++)
                for (int v = 0; v < 3; v++) {
                    const float4 q = h_tris[3 * (size_t)t + v];
                    h36[9 * (size_t)t + 3 * v] = q.x; h36[9 * (size_t)t + 3 * v + 1] = q.y; h36[9 * (size_t)t + 3 * v + 2] = q.z;
                }
            TRY(dev_alloc(r, &r->d_tris36, h36.size()));
            HIP_TRY(r, hipMemcpy(r->d_tris36, h36.data(), h36.size() * sizeof(float), hipMemcpyHostToDevice));
            r->wide.tris36 = r->d_tris36;
        }
        TRY(dev_alloc(r, &r->d_wide, h_wide.size()));
        HIP_TRY(r, hipMemcpy(r->d_wide, h_wide.data(), h_wide.size() * sizeof(float4), hipMemcpyHostToDevice));
        // deepest wide stack of THIS tree (see ensure_wide_overflow); a new scene may need a different size
        int max_pending = 0;
        for (int i = 0; i < n_boxes; i++) max_pending = std::max(max_pending, pending[i]);
        const int entries = std::min((int)WIDE_STACK_OVERFLOW, 2 * max_pending + 4);
        if (entries != r->wide_ovf_entries) { dev_free(r, r->d_wide_ovf); r->wide_ovf_entries = entries; }
        r->wide.nodes = r->d_wide; r->wide.tris = r->d_tris;
        r->wide.root_lo = make_float4(boxes[0].min[0], boxes[0].min[1], boxes[0].min[2], 0.0f);
        r->wide.root_hi = make_float4(boxes[0].max[0], boxes[0].max[1], boxes[0].max[2], 0.0f);
        // The nearest-first walk (ORDER) settles exact-t ties the way the reference does -- the triangle it meets FIRST wins,
        // trace.metal:170 -- from each triangle's position in the reference's visit order (child left+1 first, a leaf's triangles in
        // index order).  np_flatten_bvh numbers leaves breadth-first, so the index alone does not say it.  Read on ties only.
        {
            // [0]: "nothing held" (best.tri = -1) ranks before everything; [1 + t]: triangle t; [1 + n_tris]: behind the last triangle
            std::vector<int> h_rank((size_t)n_tris + 2, 0x7fffffff), st{0};
            h_rank[0] = (int)0x80000000;
            int next_rank = 0;
            while (!st.empty()) {
                const int x = st.back(); st.pop_back();
                const BoxRec& bx = boxes[x];
                if (bx.right == 0) { st.push_back(bx.left); st.push_back(bx.left + 1); continue; }   // left+1 on top: popped first
                for (int t = bx.left; t < bx.right; t++) if (h_rank[1 + (size_t)t] == 0x7fffffff) h_rank[1 + (size_t)t] = next_rank++;
            }
            TRY(dev_alloc(r, &r->d_tri_rank, h_rank.size()));
            HIP_TRY(r, hipMemcpy(r->d_tri_rank, h_rank.data(), h_rank.size() * sizeof(int), hipMemcpyHostToDevice));
            r->wide.tri_rank = r->d_tri_rank + 1;
        }
        r->n_wide = n_wide;
    }
    r->bvh.nodes = r->d_nodes; r->bvh.tris = r->d_tris;
    r->bvh.n_nodes = n_records; r->bvh.n_tris = n_tris;
    r->bvh.n_lds_nodes = std::min(n_records, LDS_NODE_CAP);
    r->bvh.lds_tris = n_tris <= LDS_TRI_CAP ? 1 : 0;
    dev_free(r, r->d_fast);
    r->bvh.fast_nodes = nullptr; r->bvh.n_fast_nodes = 0;
    if (n_fast > 0) {
        TRY(dev_alloc(r, &r->d_fast, h_fast.size()));
        HIP_TRY(r, hipMemcpy(r->d_fast, h_fast.data(), h_fast.size() * sizeof(float4), hipMemcpyHostToDevice));
        r->bvh.fast_nodes = r->d_fast; r->bvh.n_fast_nodes = ((r->debug_flags >> 7) & 1) ? 0 : n_fast;
    }
    r->n_fast = n_fast;
    // a table without inner records, every skip link pointing at the next record: all rays visit the same records in the
    // same order and the walk's control flow can be wave-uniform (closest_hit_flat)
    r->fast_flat = n_fast > 0 ? 1 : 0;
    for (int j = 0; j < n_fast; j++) {
        int skip, info;
        std::memcpy(&skip, &h_fast[2 * (size_t)j].w, 4);
        std::memcpy(&info, &h_fast[2 * (size_t)j + 1].w, 4);
        if (info < 0 || skip != j + 1) r->fast_flat = 0;
    }
    r->bvh.fast_flat = ((r->debug_flags >> 11) & 1) ? 0 : r->fast_flat;
    r->n_mats = n_mats; r->light_count = light_count; r->cam = cam;
    r->cam_tris = cam_tris;
    r->n_top = n_top;
    r->scene_ok = true;
    r->paths_share = 0;                  // re-tune the stage shares for the new scene
    r->levels_auto = 0;
    return CL2_OK;
}

int cl2_set_seeds(cl2_renderer* r, const uint32_t* seeds, size_t n_words) {
    if (!r || !seeds) return CL2_E_INVALID;
    if (n_words != 2 * (size_t)r->B) return fail(r, CL2_E_INVALID, "seed buffer must hold 2 words per pixel and sample stream (stream-major)");
    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipMemcpy(r->d_seeds, seeds, n_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    return CL2_OK;
}

int cl2_get_seeds(cl2_renderer* r, uint32_t* seeds, size_t n_words) {
    if (!r || !seeds) return CL2_E_INVALID;
    if (n_words != 2 * (size_t)r->B) return fail(r, CL2_E_INVALID, "seed buffer must hold 2 words per pixel and sample stream (stream-major)");
    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipMemcpy(seeds, r->d_seeds, n_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CL2_OK;
}

/* K independent samples of the frame per pass.  Stream k has its own seed words [2*k*W*H, 2*(k+1)*W*H) of
 * the seed buffer and is exactly what a Renderer with that seed buffer -- rank k of the sample split -- would render; the
 * accumulators receive the streams' samples in stream order.  Every launch then covers K x W x H entries: a per-level subpath
 * launch of a 1080p frame is 2 M rays on 524 k resident lanes and all tail; with K = 4 it is the 8.3 M rays of a 4K frame.
 * Frees and re-allocates the per-pixel state (seeds back to 1: set them again), keeps scene, accumulators and counters. */
int cl2_set_sample_streams(cl2_renderer* r, int streams) {
    if (!r) return CL2_E_INVALID;
    if (streams < 1 || (long long)streams * r->FB >= (1ll << TAG_PID_BITS))
        return fail(r, CL2_E_INVALID, "sample streams: need streams >= 1 and streams * W * H < 2^26 (pixel-entry bits of a connection tag)");
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    if (streams == r->streams) return CL2_OK;
    const int before = r->streams;
    free_pixel_state(r);
    r->streams = streams; r->B = streams * r->FB; r->export_stream = 0;
    r->paths_share = 0; r->levels_auto = 0;                       // launch sizes changed: measure the organisation again
    int rc = alloc_pixel_state(r);
    if (rc != CL2_OK) {
        // not enough device memory for that many streams (about 3.4 KB per entry): the handle goes back to what it had
        const std::string why = r->err;
        free_pixel_state(r);
        r->streams = before; r->B = before * r->FB;
        if (alloc_pixel_state(r) != CL2_OK) { r->scene_ok = false; return fail(r, rc, why + "; and the previous stream count could not be restored: the handle is unusable"); }
        return fail(r, rc, why + " (sample streams unchanged; seeds were reset)");
    }
    return CL2_OK;
}
int cl2_get_sample_streams(const cl2_renderer* r) { return r ? r->streams : CL2_E_INVALID; }
/* the stream that cl2_export_* / cl2_import_sample_images address (one frame's worth of records each) */
int cl2_set_export_stream(cl2_renderer* r, int stream) {
    if (!r) return CL2_E_INVALID;
    if (stream < 0 || stream >= r->streams) return fail(r, CL2_E_INVALID, "export stream out of range");
    r->export_stream = stream;
    return CL2_OK;
}

int cl2_make_light_rays(cl2_renderer* r) { STAGE_PROLOGUE(r); TRY(launch_generate(r, CL2_LIGHT, r->stream, r->sets[r->cur])); return drain(r); }
int cl2_make_camera_rays(cl2_renderer* r) { STAGE_PROLOGUE(r); TRY(launch_generate(r, CL2_CAMERA, r->stream, r->sets[r->cur])); return drain(r); }
int cl2_trace_light_rays(cl2_renderer* r) { STAGE_PROLOGUE(r); TRY(launch_subpaths(r, r->stream, r->sets[r->cur], 1)); return drain(r); }
int cl2_trace_camera_rays(cl2_renderer* r) { STAGE_PROLOGUE(r); TRY(launch_subpaths(r, r->stream, r->sets[r->cur], 2)); return drain(r); }
int cl2_join_paths(cl2_renderer* r) {
    STAGE_PROLOGUE(r);
    TRY(launch_connect(r, r->stream, r->sets[r->cur], 0));
    TRY(launch_resolve(r, r->stream, r->sets[r->cur], 0));
    return drain(r);
}
int cl2_finalize_samples(cl2_renderer* r) { STAGE_PROLOGUE(r); TRY(launch_finalize(r, r->stream)); return drain(r); }
/* The t=1 splats were added to the light image by join_paths (float atomics); nothing is left of the
 * reference's sort + bincount + gather (src/renderer.py:212-250) but a synchronisation point. */
int cl2_gather_light_image(cl2_renderer* r) { STAGE_PROLOGUE(r); return drain(r); }
int cl2_process_images(cl2_renderer* r) {
    STAGE_PROLOGUE(r);
    TRY(launch_accumulate(r, r->stream));
    r->samples += (uint64_t)r->streams;
    return drain(r);
}

/* n x run_sample as a three-stage pipeline over samples.  The only state a sample hands to the next
 * one is the seed buffer, and only the subpath stage (K1, K2, K3) touches it.  So with three subpath
 * buffer sets and two {hit, mask} sets
 *     stream       : subpaths of sample i+2
 *     stream_conn  : connection set-up + connection rays of sample i+1
 *     stream_res   : resolve, K6, accumulation of sample i
 * run side by side.  Every launch ends in a tail of a few long rays or late workgroups, the
 * streaming kernels leave the VALUs idle and the resolve kernel the memory system: the other stages'
 * workgroups fill what is left.  Each kernel sees exactly the inputs it would see in the serial
 * order (sums into the accumulators stay in sample order on stream_res), so results do not change. */
namespace {
// rays traced by the subpath kernels so far (per-workgroup slots, summed here)
int subpath_ray_tally(cl2_renderer* r, unsigned long long* out) {
    TRY(drain(r));
    std::vector<unsigned long long> slots((size_t)grid_for(r->B) * 4);
    HIP_TRY(r, hipMemcpy(slots.data(), r->d_block_stats, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t = 0;
    for (size_t b = 0; b < slots.size(); b += 4) t += slots[b];
    *out = t;
    return CL2_OK;
}

// `count` samples, serial (one stream, the current buffer set) or pipelined (rotating sets, starting with
// the current one); ends with everything complete and `cur` = the set of the last sample.
// `steady`: r->tune_period_ms = the time from the end of the first sample to the end of the last one, per sample -- what a long run
// costs per sample, without the fill of the pipeline (the first sample's subpath stage has nothing beside it).
int run_chunk(cl2_renderer* r, bool pipe, int count, bool steady = false) {
    const int first_set = r->cur;
    r->pipe_active = pipe;
    for (int i = 0; i < count; i++) {
        const int ps = pipe ? (first_set + i) % 3 : r->cur, cs = pipe ? (i & 1) : 0;
        const PathBufs* set = r->sets[ps];
        hipStream_t s_conn = pipe ? r->stream_conn : r->stream;
        hipStream_t s_res = pipe ? (pipeline_stages(r) == 2 ? r->stream_res : r->stream_conn) : r->stream;
        // resolve of sample i-3 was the last reader of this subpath set
        if (pipe && i >= 3) HIP_TRY(r, hipStreamWaitEvent(r->stream, r->ev_res[(i - 3) % 6], 0));
        TRY(launch_generate_both(r, r->stream, set));
        TRY(launch_subpaths(r, r->stream, set, 3));
        if (pipe) {
            HIP_TRY(r, hipEventRecord(r->ev_paths[i % 3], r->stream));
            HIP_TRY(r, hipStreamWaitEvent(s_conn, r->ev_paths[i % 3], 0));
            // resolve of sample i-2 was the last reader of this {hit, mask} set
            if (i >= 2) HIP_TRY(r, hipStreamWaitEvent(s_conn, r->ev_res[(i - 2) % 6], 0));
        }
        TRY(launch_connect(r, s_conn, set, cs));
        if (pipe) {
            HIP_TRY(r, hipEventRecord(r->ev_conn[i & 1], s_conn));
            HIP_TRY(r, hipStreamWaitEvent(s_res, r->ev_conn[i & 1], 0));
        }
        TRY(launch_resolve(r, s_res, set, cs));
        TRY(launch_finalize_accumulate(r, s_res));
        if (steady && count >= 2 && (i == 0 || i == count - 1)) HIP_TRY(r, hipEventRecord(r->ev_tune[i == 0 ? 0 : 1], s_res));
        if (pipe) HIP_TRY(r, hipEventRecord(r->ev_res[i % 6], s_res));
        r->samples += (uint64_t)r->streams;
        // bound the number of in-flight event pairs while profiling
        if (r->profiling && r->spans.size() > 4096) TRY(drain(r));
    }
    r->pipe_active = false;
    TRY(drain(r));
    if (steady && count >= 2) {
        float ms = 0.0f;
        HIP_TRY(r, hipEventElapsedTime(&ms, r->ev_tune[0], r->ev_tune[1]));
        r->tune_period_ms = (double)ms / (count - 1);
    }
    if (pipe && count > 0) r->cur = (first_set + count - 1) % 3;
    return CL2_OK;
}
}  // namespace

namespace {
// The automatic choices that need measurements of the scene.  Both render real samples (every candidate
// organisation gives identical results), `done` counts them.
//
// Small scenes, levels_per_launch = 0: one whole-subpath launch keeps a wave busy as long as its
// longest path lives.  In a closed scene that is every path (6.0 rays per subpath: nothing to gain,
// compaction costs 35 %); in an open scene most paths leave after a bounce or two and a wave idles on
// its last survivor.  The first sample of a scene tells which: below 4 rays per subpath the walk is cut
// into launches of 2 bounces with the survivors compacted in between (open test scene: 2.02 -> 1.62 ms).
int tune_levels(cl2_renderer* r, int& done) {
    if (split_paths(r) || r->levels_per_launch != 0 || r->levels_auto != 0) return CL2_OK;
    unsigned long long before = 0, after = 0;
    TRY(subpath_ray_tally(r, &before));
    TRY(run_chunk(r, false, 1));
    TRY(subpath_ray_tally(r, &after));
    done += 1;
    r->levels_auto = (double)(after - before) < 4.0 * 2.0 * (double)r->B ? 2 : (int)MAX_VERTS;
    return CL2_OK;
}
// Large scenes: how the two stages share the machine while the pipeline runs.  Persistent launches hold
// their wave slots until they run dry, so the stages get fixed shares (persistent_grid_paths/_conn) -- 3, 4
// or 5 eighths for the subpath stage, no fixed shares, or the serial order.  The best choice depends on the
// scene and the frame size; every candidate renders TUNE_SAMPLES samples and is timed on the host.
constexpr int TUNE_TOTAL = 9 * TUNE_SAMPLES;     // five candidates once, the best two twice more
inline bool shares_untuned(const cl2_renderer* r) { return pipeline_stages(r) != 0 && split_conn(r) && r->paths_share == 0; }
int tune_shares(cl2_renderer* r, int& done) {
    if (!shares_untuned(r)) return CL2_OK;
    const int cand[5] = {3, 4, 5, 8, SHARE_SERIAL};
    double t_of[5];
    auto time_one = [&](int e, double& t) -> int {
        // Timed on the device from the end of the candidate's first sample to the end of its last: the host clock around six
        // samples charged a pipelined organisation its fill (about one sample in six) and the serial order nothing -- round 5: with
        // one sample stream the tuner took the serial order (8.35 / 10.3 ms per sample on the 5k- / 82k-triangle scene) although the
        // pipelined one runs at 8.0 / 9.9 in a long call.
        r->paths_share = e;
        TRY(run_chunk(r, e != SHARE_SERIAL, TUNE_SAMPLES, true));
        t = r->tune_period_ms;
        done += TUNE_SAMPLES;
        return CL2_OK;
    };
    for (int k = 0; k < 5; k++) TRY(time_one(cand[k], t_of[k]));
    // the two best twice more, in turn: neighbouring organisations differ by a few per cent and one timing of six samples by about
    // as much (round 5: on the 5k-triangle scene with one sample stream the pipelined order, 7.99 ms per sample, lost the run-off
    // against the serial order, 8.35, in four tunings out of five when each was timed once more)
    int a = 0, b = 1;
    if (t_of[b] < t_of[a]) std::swap(a, b);
    for (int k = 2; k < 5; k++) {
        if (t_of[k] < t_of[a]) { b = a; a = k; }
        else if (t_of[k] < t_of[b]) b = k;
    }
    double ta = 0, tb = 0;
    for (int round = 0; round < 2; round++) {
        double t1 = 0, t2 = 0;
        TRY(time_one(cand[a], t1));
        TRY(time_one(cand[b], t2));
        ta += t1; tb += t2;
    }
    r->paths_share = (t_of[a] + ta <= t_of[b] + tb) ? cand[a] : cand[b];
    return CL2_OK;
}
}  // namespace

int cl2_run_samples(cl2_renderer* r, int n) {
    STAGE_PROLOGUE(r);
    if (n < 0) return fail(r, CL2_E_INVALID, "negative sample count");
    bool pipe = pipeline_stages(r) != 0 && n > 1;
    int done = 0;
    if (n >= 2) TRY(tune_levels(r, done));
    // the share tuner runs inside the first LONG call of a scene (its 54 samples are part of the call's n) unless
    // cl2_tune() ran it before
    if (pipe && n - done >= TUNE_TOTAL + TUNE_SAMPLES + 6) TRY(tune_shares(r, done));
    if (pipe && split_conn(r) && r->paths_share == SHARE_SERIAL) pipe = false;
    return run_chunk(r, pipe, n - done);
}

/* Makes the measured choices of the launch organisation NOW instead of inside the first (long) cl2_run_samples
 * call: the levels-per-launch probe of a small scene (1 sample) and the stage-share tuner of a large one (54
 * samples: TUNE_TOTAL).  The samples are real ones -- they advance the seeds and add to the accumulators exactly as the same
 * number of run_sample iterations would -- and *samples_rendered says how many there were.  A benchmark calls
 * this in its warm-up so that no timing experiment runs inside its clock. */
int cl2_tune(cl2_renderer* r, int* samples_rendered) {
    STAGE_PROLOGUE(r);
    int done = 0;
    TRY(tune_levels(r, done));
    TRY(tune_shares(r, done));
    if (samples_rendered) *samples_rendered = done;
    return CL2_OK;
}

int cl2_set_pipelining(cl2_renderer* r, int on) {
    if (!r) return CL2_E_INVALID;
    if (on < -1 || on > 2) return fail(r, CL2_E_INVALID, "pipelining must be -1..2");
    r->pipelining = on;
    return CL2_OK;
}

int cl2_read_accumulators(cl2_renderer* r, float* img, float* wts, int32_t* counts, float* uni, size_t n_pixels) {
    if (!r) return CL2_E_INVALID;
    if (n_pixels != (size_t)r->FB) return fail(r, CL2_E_INVALID, "n_pixels must equal W*H");
    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    const size_t B = r->FB;
    std::vector<float> h(8 * B);
    HIP_TRY(r, hipMemcpy(h.data(), r->d_acc, 8 * B * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t p = 0; p < B; p++) {
        if (img) { img[3 * p] = h[p]; img[3 * p + 1] = h[B + p]; img[3 * p + 2] = h[2 * B + p]; }
        if (wts) wts[p] = h[3 * B + p];
        if (uni) { uni[3 * p] = h[4 * B + p]; uni[3 * p + 1] = h[5 * B + p]; uni[3 * p + 2] = h[6 * B + p]; }
        if (counts) counts[p] = (int32_t)std::lrintf(h[7 * B + p]);
    }
    return CL2_OK;
}

int cl2_reset_accumulators(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    HIP_TRY(r, hipMemset(r->d_acc, 0, 8 * (size_t)r->FB * sizeof(float)));
    r->samples = 0;
    return CL2_OK;
}

static int acc_copy(cl2_renderer* r, void* dst, const void* src, size_t n_floats, hipMemcpyKind kind) {
    if (!r || !dst || !src) return CL2_E_INVALID;
    if (n_floats != 8 * (size_t)r->FB) return fail(r, CL2_E_INVALID, "packed accumulators hold 8*W*H floats");
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    HIP_TRY(r, hipMemcpy(dst, src, n_floats * sizeof(float), kind));
    return CL2_OK;
}
int cl2_read_accumulators_packed(cl2_renderer* r, float* dst, size_t n) { return acc_copy(r, dst, r ? r->d_acc : nullptr, n, hipMemcpyDeviceToHost); }
int cl2_write_accumulators_packed(cl2_renderer* r, const float* src, size_t n) { return acc_copy(r, r ? r->d_acc : nullptr, src, n, hipMemcpyHostToDevice); }

// ---------------------------------------------------------------- output stage: tone map on the device (csrc/tonemap.hpp)
namespace {
int tone_buffers(cl2_renderer* r) {
    if (!r->d_tone_partial) TRY(dev_alloc(r, &r->d_tone_partial, (size_t)TONE_BLOCKS + 1));
    if (!r->d_tone_out) TRY(dev_alloc(r, &r->d_tone_out, (size_t)3 * r->FB));
    return CL2_OK;
}
}  // namespace

int cl2_tone_log_sum(cl2_renderer* r, int which, double* sum_out) {
    if (!r || !sum_out) return CL2_E_INVALID;
    if (which < 0 || which > 2) return fail(r, CL2_E_INVALID, "picture: 0 image, 1 unweighted_image, 2 unidirectional_image");
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    TRY(tone_buffers(r));
    const int grid = std::min(grid_for(r->FB), TONE_BLOCKS);
    hipLaunchKernelGGL(k_tone_logsum, dim3(grid), dim3(256), 0, r->stream, r->d_acc, r->FB, which, r->d_tone_partial);
    hipLaunchKernelGGL(k_tone_logsum_final, dim3(1), dim3(256), 0, r->stream, r->d_tone_partial, grid, r->d_tone_partial + TONE_BLOCKS);
    HIP_TRY(r, hipGetLastError());
    HIP_TRY(r, hipMemcpyAsync(sum_out, r->d_tone_partial + TONE_BLOCKS, sizeof(double), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    return CL2_OK;
}

int cl2_tone_map(cl2_renderer* r, int which, double exposure, double white_point, double log_average, uint8_t* out_bgr, size_t n_bytes) {
    if (!r || !out_bgr) return CL2_E_INVALID;
    if (which < 0 || which > 2) return fail(r, CL2_E_INVALID, "picture: 0 image, 1 unweighted_image, 2 unidirectional_image");
    if (n_bytes != (size_t)3 * r->FB) return fail(r, CL2_E_INVALID, "the tone-mapped picture holds 3*W*H bytes");
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    TRY(tone_buffers(r));
    hipLaunchKernelGGL(k_tone_apply, dim3(grid_for(r->FB)), dim3(256), 0, r->stream, r->d_acc, r->FB, which, exposure, white_point * white_point,
                       log_average, r->d_tone_out);
    HIP_TRY(r, hipGetLastError());
    HIP_TRY(r, hipMemcpyAsync(out_bgr, r->d_tone_out, n_bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    return CL2_OK;
}

// ---------------------------------------------------------------- multi-GPU: RCCL behind the C ABI
namespace {
constexpr int COMM_SCRATCH = 16;
#define RCCL_TRY(r, api, expr)                                                                    \
    do {                                                                                          \
        ncclResult_t e_ = (expr);                                                                 \
        if (e_ != ncclSuccess) {                                                                  \
            (r)->err = std::string(#expr) + ": " + (api)->GetErrorString(e_);                     \
            return CL2_E_COMM;                                                                    \
        }                                                                                         \
    } while (0)
}  // namespace

int cl2_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int cl2_synchronize(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    HIP_TRY(r, hipDeviceSynchronize());
    return CL2_OK;
}

int cl2_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int cl2_comm_get_unique_id(void* out, size_t n_bytes) {
    if (!out || n_bytes != sizeof(ncclUniqueId)) { g_create_error = "cl2_comm_get_unique_id: the id is cl2_comm_unique_id_bytes() long"; return CL2_E_INVALID; }
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) { g_create_error = why; return CL2_E_COMM; }
    ncclUniqueId id;
    const ncclResult_t e = api->GetUniqueId(&id);
    if (e != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(e); return CL2_E_COMM; }
    std::memcpy(out, &id, sizeof id);
    return CL2_OK;
}

int cl2_comm_init_rank(cl2_renderer* r, int nranks, int rank, const void* unique_id, size_t n_bytes) {
    if (!r) return CL2_E_INVALID;
    if (!unique_id || n_bytes != sizeof(ncclUniqueId)) return fail(r, CL2_E_INVALID, "unique id must be cl2_comm_unique_id_bytes() long");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(r, CL2_E_INVALID, "need 0 <= rank < nranks");
    if (r->comm) return fail(r, CL2_E_STATE, "this renderer already has a communicator (cl2_comm_destroy first)");
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return fail(r, CL2_E_COMM, why);
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    if (!r->d_comm_scratch) TRY(dev_alloc(r, &r->d_comm_scratch, (size_t)COMM_SCRATCH));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    RCCL_TRY(r, api, api->CommInitRank(&comm, nranks, id, rank));
    r->comm = comm; r->comm_rank = rank; r->comm_nranks = nranks;
    r->comm_poisoned = false;
    return CL2_OK;
}

namespace {
// Deadline for one collective (seconds): CLIVE2_COMM_TIMEOUT_S, default 300.  Ranks reach the reduce at
// different times (scene set-up, sample counts that differ by one), so the default is generous; what it
// bounds is the wait for a rank that will never come.
double comm_deadline_seconds() {
    if (const char* e = std::getenv("CLIVE2_COMM_TIMEOUT_S")) {
        char* end = nullptr;
        const double v = std::strtod(e, &end);
        if (end != e && v > 0.0) return v;
    }
    return 300.0;
}

// Tear the communicator down without waiting for anybody: ncclCommAbort ends kernels of it that are still
// spinning on a peer and frees its resources.  Used after a failed / timed-out collective and whenever the
// handle goes away in a failed state -- a clean ncclCommDestroy would itself wait for outstanding work.
void comm_abort(cl2_renderer* r, RcclApi* api) {
    if (!r->comm) return;
    ncclComm_t comm = r->comm;
    r->comm = nullptr; r->comm_nranks = 0; r->comm_rank = 0;
    if (api && api->CommAbort) (void)api->CommAbort(comm);
}

// Wait for the collective just enqueued on r->stream: polls the stream and the communicator's asynchronous
// error state under the deadline (comm_wait.hpp).  On any failure the communicator is aborted and the
// handle keeps the reason; the caller returns CL2_E_COMM and the process is expected to exit non-zero.
int comm_wait(cl2_renderer* r, RcclApi* api, const char* what) {
    int detail = 0;
    const double deadline = comm_deadline_seconds();
    const WaitResult res = wait_collective(
        [&]() -> int {
            const hipError_t e = hipStreamQuery(r->stream);
            return e == hipSuccess ? 0 : (e == hipErrorNotReady ? 1 : -(int)e);
        },
        [&]() -> int {
            ncclResult_t ae = ncclSuccess;
            if (!api->CommGetAsyncError || api->CommGetAsyncError(r->comm, &ae) != ncclSuccess) return 0;
            return (ae == ncclSuccess || ae == ncclInProgress) ? 0 : (int)ae;
        },
        deadline, &detail);
    if (res == WAIT_DONE) return CL2_OK;
    std::string why = std::string(what) + ": ";
    if (res == WAIT_TIMEOUT) why += "not complete after " + std::to_string((int)deadline) + " s (a rank never joined the collective?); communicator aborted";
    else if (res == WAIT_ASYNC_ERROR) why += std::string("asynchronous RCCL error: ") + api->GetErrorString((ncclResult_t)detail) + "; communicator aborted";
    else why += std::string("stream error: ") + hipGetErrorString((hipError_t)(-detail)) + "; communicator aborted";
    r->comm_poisoned = true;
    comm_abort(r, api);
    r->err = why;
    return CL2_E_COMM;
}

// every failure between "communicator exists" and "collective complete" poisons the communicator: the peers are
// (or will be) inside the collective and this rank will not complete it
#define COMM_HIP_TRY(r, api, expr)                                                                 \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (r)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            (r)->comm_poisoned = true; comm_abort((r), (api));                                    \
            return CL2_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)
#define COMM_RCCL_TRY(r, api, expr)                                                                \
    do {                                                                                          \
        ncclResult_t e_ = (expr);                                                                 \
        if (e_ != ncclSuccess) {                                                                  \
            (r)->err = std::string(#expr) + ": " + (api)->GetErrorString(e_);                     \
            (r)->comm_poisoned = true; comm_abort((r), (api));                                    \
            return CL2_E_COMM;                                                                    \
        }                                                                                         \
    } while (0)
}  // namespace

/* The collective of SURVEY.md 8e: in place, on the stream every accumulating kernel ran on (drained
 * first: with the sample pipeline the accumulation runs on stream_res).  The wait for it has a deadline
 * (comm_wait): a rank that never joins makes this call fail with CL2_E_COMM instead of blocking for ever. */
int cl2_reduce_accumulators(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    if (!r->comm) return fail(r, CL2_E_STATE, r->comm_poisoned ? "the communicator was aborted after a failed collective" : "no communicator (call cl2_comm_init_rank first)");
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return fail(r, CL2_E_COMM, why);
    COMM_HIP_TRY(r, api, hipSetDevice(r->device));
    {
        const int rc = drain(r);
        if (rc != CL2_OK) { r->comm_poisoned = true; comm_abort(r, api); return rc; }
    }
    COMM_RCCL_TRY(r, api, api->AllReduce(r->d_acc, r->d_acc, 8 * (size_t)r->FB, ncclFloat, ncclSum, r->comm, r->stream));
    return comm_wait(r, api, "cl2_reduce_accumulators");
}

/* n <= 16 host doubles, summed (op 0) or maximised (op 1) over the ranks, in place: barrier, the
 * max-over-ranks clock and the whole-job ray tally of bench.py / render.py. */
int cl2_comm_allreduce_f64(cl2_renderer* r, double* values, int n, int op) {
    if (!r || !values) return CL2_E_INVALID;
    if (n < 1 || n > COMM_SCRATCH || (op != 0 && op != 1)) return fail(r, CL2_E_INVALID, "allreduce_f64: 1..16 values, op 0 (sum) or 1 (max)");
    if (!r->comm) return fail(r, CL2_E_STATE, r->comm_poisoned ? "the communicator was aborted after a failed collective" : "no communicator (call cl2_comm_init_rank first)");
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return fail(r, CL2_E_COMM, why);
    COMM_HIP_TRY(r, api, hipSetDevice(r->device));
    {
        const int rc = drain(r);
        if (rc != CL2_OK) { r->comm_poisoned = true; comm_abort(r, api); return rc; }
    }
    COMM_HIP_TRY(r, api, hipMemcpyAsync(r->d_comm_scratch, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, r->stream));
    COMM_RCCL_TRY(r, api, api->AllReduce(r->d_comm_scratch, r->d_comm_scratch, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, r->comm, r->stream));
    TRY(comm_wait(r, api, "cl2_comm_allreduce_f64"));
    HIP_TRY(r, hipMemcpy(values, r->d_comm_scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return CL2_OK;
}

/* What the COMMUNICATOR says about itself (not what the launcher's environment says): ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice, and the PCI address of this rank's GPU as one integer, domain << 16 | bus << 8 | device << 3 | function
 * (hipDeviceGetPCIBusId).  A job line that lists nranks distinct addresses shows that RCCL saw that many GPUs. */
int cl2_comm_info(cl2_renderer* r, cl2_comm_info_t* out) {
    if (!r || !out) return CL2_E_INVALID;
    std::memset(out, 0, sizeof *out);
    out->device_ordinal = r->device;
    HIP_TRY(r, hipSetDevice(r->device));
    char bus[32] = {0};
    HIP_TRY(r, hipDeviceGetPCIBusId(bus, (int)sizeof bus, r->device));
    unsigned dom = 0, b = 0, dev = 0, fn = 0;
    if (std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &dev, &fn) >= 3)
        out->pci_address = ((int64_t)dom << 16) | ((int64_t)b << 8) | ((int64_t)dev << 3) | (int64_t)fn;
    std::snprintf(out->pci_bus_id, sizeof out->pci_bus_id, "%s", bus);
    if (!r->comm) return CL2_OK;                       // nranks = 0: no communicator
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return fail(r, CL2_E_COMM, why);
    int n = 0, rank = -1, cudev = -1;
    RCCL_TRY(r, api, api->CommCount(r->comm, &n));
    RCCL_TRY(r, api, api->CommUserRank(r->comm, &rank));
    RCCL_TRY(r, api, api->CommCuDevice(r->comm, &cudev));
    out->nranks = n; out->rank = rank; out->comm_device = cudev;
    return CL2_OK;
}

/* The host's way of saying "this rank failed, do not wait for anybody": the communicator is torn down with
 * ncclCommAbort at once, so peers blocked in a collective with this rank see an error or their own deadline
 * instead of waiting for a rank that is about to exit.  Harmless without a communicator. */
int cl2_comm_abort(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    if (!r->comm) return CL2_OK;
    std::string why;
    RcclApi* api = rccl_api(why);
    (void)hipSetDevice(r->device);
    r->comm_poisoned = true;
    comm_abort(r, api);
    return CL2_OK;
}

int cl2_comm_destroy(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    if (!r->comm) return CL2_OK;
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return fail(r, CL2_E_COMM, why);
    (void)hipSetDevice(r->device);
    // a handle whose own streams are in an error state, or whose last collective failed, must not wait in
    // ncclCommDestroy for work that will never finish
    if (r->comm_poisoned || drain(r) != CL2_OK) { comm_abort(r, api); return CL2_OK; }
    ncclComm_t comm = r->comm;
    r->comm = nullptr; r->comm_nranks = 0; r->comm_rank = 0;
    RCCL_TRY(r, api, api->CommDestroy(comm));
    return CL2_OK;
}

int cl2_set_profiling(cl2_renderer* r, int level) {
    if (!r) return CL2_E_INVALID;
    r->profiling = level < 0 ? 0 : (level > 2 ? 2 : level);
    return CL2_OK;
}
int cl2_set_levels_per_launch(cl2_renderer* r, int levels) {
    if (!r) return CL2_E_INVALID;
    if (levels < 0 || levels > MAX_VERTS) return fail(r, CL2_E_INVALID, "levels_per_launch must be 0 (by survival) or 1..6");
    r->levels_per_launch = levels;
    return CL2_OK;
}
int cl2_selftest_exact_math(cl2_renderer* r, uint64_t* rcp_mismatches, uint64_t* divpi_mismatches) {
    if (!r || !rcp_mismatches || !divpi_mismatches) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    unsigned long long* d = nullptr;
    TRY(dev_alloc(r, &d, (size_t)2));
    int rc = CL2_OK;
    if (hipMemset(d, 0, 16) != hipSuccess) rc = fail(r, CL2_E_HIP, "selftest memset failed");
    if (rc == CL2_OK) {
        hipLaunchKernelGGL(k_selftest_exact_math, dim3(4096), dim3(BLOCK), 0, r->stream, d);
        rc = drain(r);
    }
    unsigned long long h[2] = {0, 0};
    if (rc == CL2_OK && hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "selftest copy failed");
    dev_free(r, d);
    *rcp_mismatches = h[0]; *divpi_mismatches = h[1];
    return rc;
}
int cl2_probe_math(cl2_renderer* r, int which, const float* in, size_t n, float* out) {
    if (!r || !in || !out || which < 0 || which > 7) return CL2_E_INVALID;
    if (n == 0) return CL2_OK;
    HIP_TRY(r, hipSetDevice(r->device));
    float *d_in = nullptr, *d_out = nullptr;
    int rc = dev_alloc(r, &d_in, n);
    if (rc == CL2_OK) rc = dev_alloc(r, &d_out, n);
    if (rc == CL2_OK && hipMemcpy(d_in, in, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe upload failed");
    if (rc == CL2_OK) {
        hipLaunchKernelGGL(k_probe_math, dim3(2048), dim3(BLOCK), 0, r->stream, which, n, d_in, d_out);
        rc = drain(r);
    }
    if (rc == CL2_OK && hipMemcpy(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe download failed");
    dev_free(r, d_in); dev_free(r, d_out);
    return rc;
}
int cl2_probe_bounce(cl2_renderer* r, int from_camera, const float* in, size_t n, float* out) {
    if (!r || !in || !out) return CL2_E_INVALID;
    if (n == 0) return CL2_OK;
    HIP_TRY(r, hipSetDevice(r->device));
    float *d_in = nullptr, *d_out = nullptr;
    int rc = dev_alloc(r, &d_in, 12 * n);
    if (rc == CL2_OK) rc = dev_alloc(r, &d_out, 8 * n);
    if (rc == CL2_OK && hipMemcpy(d_in, in, 12 * n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe upload failed");
    if (rc == CL2_OK) {
        hipLaunchKernelGGL(k_probe_bounce, dim3(1024), dim3(BLOCK), 0, r->stream, n, from_camera, d_in, d_out);
        rc = drain(r);
    }
    if (rc == CL2_OK && hipMemcpy(out, d_out, 8 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe download failed");
    dev_free(r, d_in); dev_free(r, d_out);
    return rc;
}
int cl2_set_traversal_mode(cl2_renderer* r, int mode) {
    if (!r) return CL2_E_INVALID;
    if (mode < 0 || mode > 5) return fail(r, CL2_E_INVALID, "traversal mode must be 0 (auto), 1 (fused), 2 (persistent, per level), 3 (fused subpaths, persistent connection rays), 4 (persistent whole subpaths) or 5 (persistent, exact 4-wide walk)");
    r->traversal_mode = mode;
    r->paths_share = 0;
    return CL2_OK;
}
int cl2_query_organisation(cl2_renderer* r, cl2_organisation* out) {
    if (!r || !out) return CL2_E_INVALID;
    if (!r->scene_ok) return fail(r, CL2_E_STATE, "no scene uploaded (call cl2_upload_scene first)");
    std::memset(out, 0, sizeof *out);
    out->tree_in_lds = tree_in_lds(r) ? 1 : 0;
    out->persistent_subpaths = split_paths(r) ? 1 : 0;
    out->persistent_connections = split_conn(r) ? 1 : 0;
    out->two_tris_per_step = two_tris_per_step(r) ? 1 : 0;
    out->n_records = r->bvh.n_nodes;
    out->n_lds_records = r->bvh.n_lds_nodes;
    out->n_top_renumbered = r->n_top;
    out->lds_triangles = r->bvh.lds_tris;
    out->levels_per_launch = split_paths(r) ? 1 : effective_levels(r);
    out->paths_share = r->paths_share;
    out->pipeline_stages = pipeline_stages(r);
    out->wide_nodes = r->n_wide;
    out->pruned_records = r->bvh.n_fast_nodes;
    out->wide_connections = (wide_walk(r) && split_conn(r)) ? 1 : 0;
    out->tree_bytes = (int64_t)r->bvh.n_nodes * 32 + (int64_t)r->bvh.n_tris * 48;
    out->sample_streams = r->streams;
    return CL2_OK;
}
/* Child order of the 4-wide walks (round 6).  0 (default) = the reference's fixed order (trace.metal:157-160): bit-exact.  1 = the
 * passing children of a node nearest first: fewer node visits and triangle tests per ray, NOT bit-exact by construction (exact-t ties
 * between two triangles, hits a few ulp in front of their leaf box; csrc/bvh_wide.hpp ORDER).  Opt-in; applies to scenes whose tree is
 * read through the caches (an LDS-resident tree such as the Cornell box has no 4-wide walk and renders the same either way); whole-
 * subpath launches give way to per-level ones, and the reference-walk tallies (cl2_set_counting(1)) stay the binary walk's. */
int cl2_set_traversal_order(cl2_renderer* r, int order) {
    if (!r) return CL2_E_INVALID;
    if (order != 0 && order != 1) return fail(r, CL2_E_INVALID, "traversal order must be 0 (reference order, exact) or 1 (nearest child first)");
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    r->traversal_order = order;
    return CL2_OK;
}
int cl2_get_traversal_order(const cl2_renderer* r) { return r ? r->traversal_order : CL2_E_INVALID; }
int cl2_set_debug_flags(cl2_renderer* r, int flags) {
    if (!r) return CL2_E_INVALID;
#ifndef CL2_TEST_VARIANT
    // bits 0-2 switch parts of the resolve stage OFF (timing dissection: the render is invalid); they exist only in
    // the test variant of the library and cannot be reached from the shipped one
    if (flags & 7) return fail(r, CL2_E_INVALID, "debug bits 0-2 (skip parts of the resolve stage: invalid renders) exist only in the test variant of the library");
#endif
    if (flags & ~CL2_DEBUG_KNOWN_BITS) return fail(r, CL2_E_INVALID, "unknown debug flag bits (see include/clive2_amd.h)");
    if (((flags >> 4) & 7) != 0 && ((flags >> 4) & 7) != 7) return fail(r, CL2_E_INVALID, "debug bits 4-6 must be 0 or 7");
    r->debug_flags = flags;
    r->bvh.n_fast_nodes = ((flags >> 7) & 1) ? 0 : r->n_fast;      // bit 7: walk the full table (A/B of the pruned one)
    r->bvh.fast_flat = ((flags >> 11) & 1) ? 0 : r->fast_flat;     // bit 11: per-lane walk of a flat pruned table (A/B of the wave-uniform one)
    return CL2_OK;
}
/* Reproducible light image (off by default): the t = 1 contributions are written as records, sorted by (target pixel, s, source
 * pixel) and summed per target in that order (det_splat.hpp) instead of being added with float atomics in whatever order the
 * hardware serves them.  Two renders of the same scene and seeds then give the same bytes in all four accumulators -- as the
 * reference's sort + bincount + gather chain does (src/renderer.py:97-111, :213-250).  Costs a radix sort of 6 x W x H keys per
 * pass and 32 bytes of record space per slot. */
int cl2_set_reproducible(cl2_renderer* r, int on) {
    if (!r) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    r->reproducible = on != 0;
    return CL2_OK;
}
int cl2_get_reproducible(const cl2_renderer* r) { return r ? (r->reproducible ? 1 : 0) : CL2_E_INVALID; }

int cl2_set_subpath_gather(cl2_renderer* r, int lanes, int wait_steps) {
    if (!r) return CL2_E_INVALID;
    if (lanes < 0 || lanes > 64 || wait_steps < 0 || wait_steps > 1000) return fail(r, CL2_E_INVALID, "subpath gather: 0..64 lanes (0 = default 32), 0..1000 steps (0 = default 48)");
    r->gather_lanes = lanes ? lanes : 32;
    r->gather_wait = wait_steps ? wait_steps : 48;
    return CL2_OK;
}
#ifdef CL2_WALK_HISTO
/* instrumentation build only: read (and clear) the pass statistics of the wide walk, 6 x 65 counters */
int cl2_walk_histo(cl2_renderer* r, unsigned long long* out) {
    if (!r || !out) return CL2_E_INVALID;
    TRY(drain(r));
    HIP_TRY(r, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_histo), sizeof(unsigned long long) * 6 * 65));
    std::vector<unsigned long long> z(6 * 65, 0ull);
    HIP_TRY(r, hipMemcpyToSymbol(HIP_SYMBOL(g_walk_histo), z.data(), sizeof(unsigned long long) * 6 * 65));
    return CL2_OK;
}
#endif
int cl2_set_counting(cl2_renderer* r, int mode) {
    if (!r) return CL2_E_INVALID;
    if (mode < 0 || mode > 2) return fail(r, CL2_E_INVALID, "counting: 0 off, 1 the reference walk's tallies (binary walk), 2 the 4-wide walk's own tallies");
    r->counting = mode;
    return CL2_OK;
}

/* cl2_set_counting(2): what the 4-wide walk itself fetched since the last cl2_reset_counters -- rays, wide nodes visited, distinct
 * triangle records read, stack entries spilled to the global overflow array, binary records visited by rays with a non-finite
 * 1/d -- separately for the subpath launches and the connection launch.  Zero where the scene does not take the wide walk. */
int cl2_read_walk_tallies(cl2_renderer* r, cl2_walk_tallies* out) {
    if (!r || !out) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    Stats s;
    HIP_TRY(r, hipMemcpy(&s, r->d_stats, sizeof s, hipMemcpyDeviceToHost));
    static_assert(sizeof(out->subpath) == sizeof(s.walk[0]) && sizeof(out->connection) == sizeof(s.walk[1]), "tally layout");
    std::memcpy(&out->subpath, s.walk[0], sizeof s.walk[0]);
    std::memcpy(&out->connection, s.walk[1], sizeof s.walk[1]);
    return CL2_OK;
}

int cl2_read_counters(cl2_renderer* r, cl2_counters* out) {
    if (!r || !out) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    Stats s;
    HIP_TRY(r, hipMemcpy(&s, r->d_stats, sizeof s, hipMemcpyDeviceToHost));
    {   // per-workgroup slots of the subpath kernel
        std::vector<unsigned long long> slots((size_t)grid_for(r->B) * 4);
        HIP_TRY(r, hipMemcpy(slots.data(), r->d_block_stats, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < slots.size(); b += 4) {
            s.rays += slots[b]; s.box_tests += slots[b + 1]; s.tri_tests += slots[b + 2]; s.counted_rays += slots[b + 3];
        }
    }
    std::memset(out, 0, sizeof *out);
    out->rays = s.rays; out->conn_rays = s.conn_rays; out->box_tests = s.box_tests; out->tri_tests = s.tri_tests;
    out->counted_rays = s.counted_rays;
    out->samples = r->samples;
    out->ms_generate = r->ms[ST_GENERATE]; out->ms_traverse_paths = r->ms[ST_TRAVERSE_PATHS]; out->ms_bounce = r->ms[ST_BOUNCE];
    out->ms_connect_setup = r->ms[ST_CONNECT_SETUP]; out->ms_traverse_conn = r->ms[ST_TRAVERSE_CONN];
    out->ms_connect_resolve = r->ms[ST_CONNECT_RESOLVE]; out->ms_finalize = r->ms[ST_FINALIZE]; out->ms_accumulate = r->ms[ST_ACCUMULATE];
    out->launches_traverse_paths = r->launches_tp; out->launches_traverse_conn = r->launches_tc;
    out->rays_traverse_paths = s.rays - s.conn_rays; out->rays_traverse_conn = s.conn_rays;
    return CL2_OK;
}

int cl2_reset_counters(cl2_renderer* r) {
    if (!r) return CL2_E_INVALID;
    HIP_TRY(r, hipSetDevice(r->device));
    TRY(drain(r));
    HIP_TRY(r, hipMemset(r->d_stats, 0, sizeof(Stats)));
    HIP_TRY(r, hipMemset(r->d_block_stats, 0, (size_t)grid_for(r->B) * 4 * sizeof(unsigned long long)));
    for (double& m : r->ms) m = 0.0;
    r->launches_tp = r->launches_tc = 0;
    return CL2_OK;
}

int cl2_export_rays(cl2_renderer* r, int which, void* out, size_t n_records) {
    STAGE_PROLOGUE(r);
    if (!out || (which != CL2_LIGHT && which != CL2_CAMERA) || n_records != (size_t)r->FB) return fail(r, CL2_E_INVALID, "bad export_rays arguments");
    RayRec* d = nullptr;
    TRY(dev_alloc(r, &d, (size_t)r->FB));
    hipLaunchKernelGGL(k_export_rays, dim3(grid_for(r->FB)), dim3(BLOCK), 0, r->stream, r->FB, r->export_stream * r->FB, r->sets[r->cur][which], which == CL2_CAMERA ? 1 : 0, d);
    int rc = drain(r);
    if (rc == CL2_OK && hipMemcpy(out, d, (size_t)r->FB * sizeof(RayRec), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "export copy failed");
    dev_free(r, d);
    return rc;
}

int cl2_export_paths(cl2_renderer* r, int which, void* out, size_t n_records) {
    STAGE_PROLOGUE(r);
    if (!out || (which != CL2_LIGHT && which != CL2_CAMERA) || n_records != (size_t)r->FB) return fail(r, CL2_E_INVALID, "bad export_paths arguments");
    unsigned char* d = nullptr;
    TRY(dev_alloc(r, &d, (size_t)r->FB * 1040));
    hipLaunchKernelGGL(k_export_paths, dim3(grid_for(r->FB)), dim3(BLOCK), 0, r->stream, r->FB, r->export_stream * r->FB, r->B, r->sets[r->cur][which], which == CL2_CAMERA ? 1 : 0, d);
    int rc = drain(r);
    if (rc == CL2_OK && hipMemcpy(out, d, (size_t)r->FB * 1040, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "export copy failed");
    dev_free(r, d);
    return rc;
}

int cl2_export_aggregators(cl2_renderer* r, void* out, size_t n_records) {
    STAGE_PROLOGUE(r);
    if (!out || n_records != (size_t)r->FB) return fail(r, CL2_E_INVALID, "bad export_aggregators arguments");
    float* d = nullptr;
    TRY(dev_alloc(r, &d, (size_t)r->FB * 32));
    hipLaunchKernelGGL(k_export_aggregators, dim3(grid_for(r->FB)), dim3(BLOCK), 0, r->stream, r->FB, r->export_stream * r->FB, r->B, r->d_agg, d);
    int rc = drain(r);
    if (rc == CL2_OK && hipMemcpy(out, d, (size_t)r->FB * 128, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "export copy failed");
    dev_free(r, d);
    return rc;
}

int cl2_export_sample_images(cl2_renderer* r, float* fin4, float* light4, float* sw, float* uni4, size_t n_pixels) {
    STAGE_PROLOGUE(r);
    if (n_pixels != (size_t)r->FB) return fail(r, CL2_E_INVALID, "n_pixels must equal W*H");
    TRY(drain(r));
    const size_t B = r->FB, off = (size_t)r->export_stream * r->FB;
    if (fin4) HIP_TRY(r, hipMemcpy(fin4, r->d_finalized + off, B * sizeof(float4), hipMemcpyDeviceToHost));
    if (light4) HIP_TRY(r, hipMemcpy(light4, r->d_light_image + off, B * sizeof(float4), hipMemcpyDeviceToHost));
    if (sw) HIP_TRY(r, hipMemcpy(sw, r->d_sample_w + off, B * sizeof(float), hipMemcpyDeviceToHost));
    if (uni4) HIP_TRY(r, hipMemcpy(uni4, r->d_uni + off, B * sizeof(float4), hipMemcpyDeviceToHost));
    return CL2_OK;
}

int cl2_import_sample_images(cl2_renderer* r, const float* fin4, const float* light4, const float* sw, const float* uni4, size_t n_pixels) {
    STAGE_PROLOGUE(r);
    if (n_pixels != (size_t)r->FB) return fail(r, CL2_E_INVALID, "n_pixels must equal W*H");
    TRY(drain(r));
    const size_t B = r->FB, off = (size_t)r->export_stream * r->FB;
    if (fin4) HIP_TRY(r, hipMemcpy(r->d_finalized + off, fin4, B * sizeof(float4), hipMemcpyHostToDevice));
    if (light4) HIP_TRY(r, hipMemcpy(r->d_light_image + off, light4, B * sizeof(float4), hipMemcpyHostToDevice));
    if (sw) HIP_TRY(r, hipMemcpy(r->d_sample_w + off, sw, B * sizeof(float), hipMemcpyHostToDevice));
    if (uni4) HIP_TRY(r, hipMemcpy(r->d_uni + off, uni4, B * sizeof(float4), hipMemcpyHostToDevice));
    return CL2_OK;
}

int cl2_probe_traverse(cl2_renderer* r, const void* rays_v, size_t n_rays, int32_t* best_i, float* best_t, float* u, float* v) {
    STAGE_PROLOGUE(r);
    if (!rays_v || !best_i || !best_t || !u || !v) return fail(r, CL2_E_INVALID, "NULL probe array");
    if (n_rays == 0) return CL2_OK;
    if (n_rays > (size_t)1 << 30) return fail(r, CL2_E_INVALID, "too many probe rays");
    const RayRec* rays = static_cast<const RayRec*>(rays_v);
    std::vector<float4> o(n_rays), d(n_rays), h(n_rays);
    for (size_t i = 0; i < n_rays; i++) {
        o[i] = make_float4(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2], 0.0f);
        d[i] = make_float4(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2], 0.0f);
    }
    float4 *d_o = nullptr, *d_d = nullptr, *d_h = nullptr;
    unsigned* d_n = nullptr;
    int rc = dev_alloc(r, &d_o, n_rays);
    if (rc == CL2_OK) rc = dev_alloc(r, &d_d, n_rays);
    if (rc == CL2_OK) rc = dev_alloc(r, &d_h, n_rays);
    if (rc == CL2_OK) rc = dev_alloc(r, &d_n, (size_t)1);
    if (rc == CL2_OK) {
        const unsigned n = (unsigned)n_rays;
        bool ok = hipMemcpy(d_o, o.data(), n_rays * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(d_d, d.data(), n_rays * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(d_n, &n, sizeof n, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) rc = fail(r, CL2_E_HIP, "probe upload failed");
    }
    if (rc == CL2_OK && r->traversal_mode == 5 && r->n_wide > 0 && !count_ref(r)) {
        // the probe through the exact 4-wide walk (rays with a non-finite 1/d take the binary walk inside it)
        if (hipMemsetAsync(r->d_work, 0, WORK_STRIDE * sizeof(unsigned), r->stream) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe memset failed");
        if (rc == CL2_OK) {
            PathRaySource src{nullptr, d_o, d_d, d_h};
            rc = launch_wide(r, r->stream, 0, d_n, r->d_work, src, 0);
            if (rc == CL2_OK) rc = drain(r);
        }
    } else if (rc == CL2_OK) {
        if (count_ref(r))
            hipLaunchKernelGGL(k_traverse_paths<true>, dim3(grid_for(n_rays)), dim3(BLOCK), bvh_lds_bytes(r), r->stream, r->bvh, (const int*)nullptr, d_n, d_o, d_d, d_h, r->d_stats);
        else
            hipLaunchKernelGGL(k_traverse_paths<false>, dim3(grid_for(n_rays)), dim3(BLOCK), bvh_lds_bytes(r), r->stream, r->bvh, (const int*)nullptr, d_n, d_o, d_d, d_h, r->d_stats);
        rc = drain(r);
    }
    if (rc == CL2_OK && hipMemcpy(h.data(), d_h, n_rays * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(r, CL2_E_HIP, "probe download failed");
    if (rc == CL2_OK)
        for (size_t i = 0; i < n_rays; i++) {
            std::memcpy(&best_i[i], &h[i].x, 4);
            best_t[i] = h[i].y; u[i] = h[i].z; v[i] = h[i].w;
        }
    dev_free(r, d_o); dev_free(r, d_d); dev_free(r, d_h); dev_free(r, d_n);
    return rc;
}

}  // extern "C"
