/* clive2_amd.h -- C ABI of the MI355X bidirectional path tracer (libclive2_amd.so).
 *
 * Drop-in boundary for the reference's Metal dispatch.  In pmclaugh/Clive2 the class
 * `Renderer` (src/renderer.py:16-352) owns ~25 `metalcompute` buffers and launches the eight
 * kernels of src/trace.metal through `dev.kernel(text).function(name)(n, *buffers)`
 * (src/renderer.py:27-37, :113-250); `create_scene` uploads nine scene buffers with
 * `dev.buffer(...)` (src/scene.py:74-89).  The entry points below are what a ctypes / cffi
 * binding of that class binds instead: plain pointers and sizes, no Python, no torch types.
 *
 *   - All record pointers use the reference's AoS layouts (src/struct_types.py:4-85):
 *     Box 48 B, Triangle 128 B, Material 48 B, Camera 112 B, Ray 128 B, Path 1040 B.
 *   - Every function returns 0 on success or a negative CL2_E_* code; cl2_last_error() returns
 *     a message for the last failure on that handle (replaces `metalcompute.error`,
 *     src/render.py:39).  No HIP failure aborts the process.
 *   - The library owns all device memory.  Host arrays passed in are copied during the call;
 *     outputs are written into caller-allocated arrays whose element counts are checked.
 *   - Calls are synchronous (they return after the stream has drained) and not re-entrant per
 *     handle; use one handle per GPU, one process (or thread) per handle.
 */
#ifndef CLIVE2_AMD_H
#define CLIVE2_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cl2_renderer cl2_renderer;

enum {
    CL2_OK = 0,
    CL2_E_INVALID = -1,      /* bad argument / scene failed validation */
    CL2_E_HIP = -2,          /* HIP runtime error (see cl2_last_error) */
    CL2_E_STATE = -3,        /* call sequence error (e.g. no scene uploaded) */
    CL2_E_NOMEM = -4,
    CL2_E_COMM = -5          /* RCCL error, or librccl could not be loaded (see cl2_last_error) */
};

/* Which subpath / ray buffer: reference `light_ray_buffer` / `camera_ray_buffer`,
 * `out_light_paths` / `out_camera_paths` (src/renderer.py:51-52, :58, :79). */
enum { CL2_LIGHT = 0, CL2_CAMERA = 1 };

/* Stage timers and tallies accumulated since cl2_reset_counters (replaces the reference's
 * `@timed` prints, src/constants.py:39-49).  One "ray" = one closest-hit BVH query =
 * one call of traverse_bvh (src/trace.metal:144).  Times are GPU milliseconds measured
 * with HIP events on the renderer's stream; they are only collected while profiling is on.
 * ms_traverse_paths = the subpath launches (closest hit + bounce per level; in the large-scene
 * organisation: the persistent traversal launches, with their bounce launches under ms_bounce). */
typedef struct {
    uint64_t rays;            /* all closest-hit queries */
    uint64_t conn_rays;       /* the subset issued by the connection stage */
    uint64_t box_tests;       /* node tests   (only counted while counting is on) */
    uint64_t tri_tests;       /* triangle tests (only counted while counting is on) */
    uint64_t counted_rays;    /* rays traced while counting was on */
    uint64_t samples;         /* completed run_sample iterations */
    double ms_generate, ms_traverse_paths, ms_bounce, ms_connect_setup, ms_traverse_conn,
           ms_connect_resolve, ms_finalize, ms_accumulate;
    uint64_t launches_traverse_paths, launches_traverse_conn;
    uint64_t rays_traverse_paths, rays_traverse_conn;   /* rays inside the timed launches */
} cl2_counters;

/* -- lifetime: replaces metalcompute.Device() + Renderer.__init__/__del__
 *    (src/scene.py:29, src/renderer.py:17-84, :318-352) -- */
int cl2_create(int device_ordinal, int pixel_width, int pixel_height, cl2_renderer** out);
void cl2_destroy(cl2_renderer* r);
const char* cl2_last_error(const cl2_renderer* r);   /* r may be NULL: error of the last failed cl2_create */
int cl2_abi_version(void);

/* -- native host BVH builder: construct_BVH + np_flatten_bvh (src/bvh.py:288-313, :329-389) in one
 *    call, O(n log n).  Inputs: per-triangle AABBs (float64, n x 3 each).  Outputs: Box records in the
 *    reference's flattened convention (capacity >= 2n-1 is always enough) and the leaf-ordered triangle
 *    permutation.  Needs no GPU.  On failure cl2_last_error(NULL) holds the message. -- */
int cl2_build_bvh(const double* tri_min, const double* tri_max, int64_t n_triangles, int max_members, int max_depth,
                  void* out_boxes, int64_t box_capacity, int64_t* n_boxes_out, int64_t* out_perm);

/* The same outputs from a GPU builder, for scenes whose set-up time matters (turntables: src/movie.py:29-55 builds one
 * scene per frame): triangles sorted along a 63-bit Morton curve of their centroids, the hierarchy above them built by
 * PLOC (every cluster merges with its nearest neighbour by union area within 8 positions along the curve, when the
 * choice is mutual; rounds until one cluster is left), subtrees of <= max_members triangles collapsed into leaves.
 * 1M triangles: 53 ms; renders a sample as fast as the reference's SAH tree (the round-2 radix tree, kept behind the
 * environment variable CLIVE2_GPU_BVH=lbvh, was 11 % slower).  Any tree in the convention renders the same picture as
 * far as the tracer is concerned; this one differs from the reference's.  The smaller child subtree is stored at left+1
 * (popped first by the traversal), which bounds the reference's traversal stack by log2(n).  Needs a GPU.  On failure
 * cl2_last_error(NULL) holds the message. */
int cl2_build_bvh_gpu(int device_ordinal, const double* tri_min, const double* tri_max, int64_t n_triangles, int max_members,
                      void* out_boxes, int64_t box_capacity, int64_t* n_boxes_out, int64_t* out_perm);
/* internal: lets the other translation units of the library leave a message for cl2_last_error(NULL) */
void cl2_set_create_error(const char* message);

/* -- scene upload: replaces the nine dev.buffer(...) uploads of create_scene
 *    (src/scene.py:74-89).  Arrays are in the reference layouts; light_* are the emitter
 *    triangle list, its areas and its indices into `triangles`. -- */
int cl2_upload_scene(cl2_renderer* r,
                     const void* boxes, int n_boxes,
                     const void* triangles, int n_triangles,
                     const void* materials, int n_materials,
                     const void* camera,
                     const void* light_triangles, const float* light_surface_areas,
                     const int32_t* light_triangle_indices, int light_count);

/* -- RNG state: the (batch,2) uint32 xorshift buffer of src/renderer.py:54,86-87 -- */
/*    n_words = 2 * streams * W * H (stream-major; see cl2_set_sample_streams) */
int cl2_set_seeds(cl2_renderer* r, const uint32_t* seeds, size_t n_words);
int cl2_get_seeds(cl2_renderer* r, uint32_t* seeds, size_t n_words);

/* -- sample streams: K independent samples of the frame per pass.  The reference's Renderer owns ONE seed buffer
 *    (src/renderer.py:54, :86-87), so K renderers -- the ranks of the sample split, SURVEY 8e -- own K; a handle with K
 *    streams holds those K buffers (seed words [2 k W H, 2 (k+1) W H) belong to stream k) and every stage call / every
 *    pass of cl2_run_samples renders one sample of EACH stream, stream k being exactly what a handle seeded with buffer k
 *    alone would render; the accumulators receive the streams' samples in stream order.  What it buys: every launch carries
 *    K x W x H work items (a per-level subpath launch of a 1080p frame is 2 M rays on 524 k resident lanes -- all tail).
 *    Default 1 = the reference's single-buffer sequence.  Needs K * W * H < 2^26.  The call frees and re-allocates the
 *    per-pixel device state (seeds return to 1; scene, accumulators and counters are kept).
 *    cl2_set_export_stream picks the stream that cl2_export_* and cl2_{ex,im}port_sample_images address. -- */
int cl2_set_sample_streams(cl2_renderer* r, int streams);
int cl2_get_sample_streams(const cl2_renderer* r);
int cl2_set_export_stream(cl2_renderer* r, int stream);

/* -- the per-sample pipeline.  The eight stage calls mirror Renderer's stage methods
 *    (src/renderer.py:113-278) for stage-level parity work; cl2_run_samples(n) is
 *    n x Renderer.run_sample (src/renderer.py:281-291) without returning to the host. -- */
int cl2_make_light_rays(cl2_renderer* r);
int cl2_make_camera_rays(cl2_renderer* r);
int cl2_trace_light_rays(cl2_renderer* r);
int cl2_trace_camera_rays(cl2_renderer* r);
int cl2_join_paths(cl2_renderer* r);
int cl2_finalize_samples(cl2_renderer* r);
int cl2_gather_light_image(cl2_renderer* r);
int cl2_process_images(cl2_renderer* r);
int cl2_run_samples(cl2_renderer* r, int n);
/* Two launch-organisation choices are MEASURED on the scene: bounces per launch for LDS-resident scenes (1 sample)
 * and the share of the machine each pipeline stage gets on large scenes (9 candidates x 6 samples, the best two twice more in turn: 54 samples, every candidate timed from device events).  By default they
 * are made inside the first cl2_run_samples call that is long enough (>= 2 / >= 66 samples); cl2_tune makes them now.
 * Its samples are real ones (seeds advance, accumulators grow, exactly as that many run_sample iterations would);
 * *samples_rendered (may be NULL) says how many.  Benchmarks call it in their warm-up.  No reference counterpart
 * (src/renderer.py:281-291 is one fixed launch sequence). */
int cl2_tune(cl2_renderer* r, int* samples_rendered);

/* Subpath levels (bounces) traced per launch: 6 walks a whole subpath in one launch with its state in
 * registers; 1 compacts the survivors after every bounce (pays when most paths die early: open
 * scenes); 0 (default) decides between 6 and 2 from the rays per subpath of the scene's first sample.
 * Results are identical for every setting. */
int cl2_set_levels_per_launch(cl2_renderer* r, int levels);
/* Traversal organisation: 1 = one ray per lane inside the subpath / connection kernels (best when the
 * tree is LDS-resident), 2 = persistent traversal launches with lane-level ray replacement + one
 * bounce launch per level (rays of very different cost: large trees), 3 = subpaths as in 1, connection
 * rays as in 2 (mid-size trees in serial order: no per-level launch tails), 4 = connection rays as in 2 and
 * both subpaths of a pixel -- light, then camera, all levels -- walked by one lane of ONE persistent launch
 * per sample, the bounces batched per wave (one launch tail instead of 24), 5 = as 2 with the connection rays walked
 * over an exact 4-wide collapse of the tree (half the dependent fetches, same decisions: csrc/bvh_wide.hpp), 0 (default)
 * = 1 for LDS-resident trees, otherwise 2 while the sample pipeline runs and 4 in serial order, with the 4-wide walk
 * for the connection rays when the tree is at most 16 MB.  Results are identical for every setting. */
int cl2_set_traversal_mode(cl2_renderer* r, int mode);
/* Sample pipeline of cl2_run_samples.  The seed buffer is the only state one sample hands to the next
 * (src/renderer.py:86-87) and only the subpath stage (K1, K2, K3) touches it, so later stages of
 * sample i can run beside the subpath stage of the following samples, on their own HIP streams and
 * buffer sets; every kernel sees the inputs of the serial order, results are the same.  The
 * reference's run_sample is strictly serial (src/renderer.py:281-291).
 *   0  serial, one stream
 *   1  two stages: subpaths of sample i+1 | connections, K6, accumulation of sample i
 *   2  three stages: subpaths of i+2 | connection set-up + connection rays of i+1 | resolve, K6,
 *      accumulation of i
 *  -1  (default) by frame size: three stages up to 2^19 pixels (small launches do not fill the
 *      machine: 256x256 runs at 10.1 / 13.9 / 18.4 Grays/s with 0 / 1 / 2), two above (equal from 1080p on) */
int cl2_set_pipelining(cl2_renderer* r, int stages);

/* -- accumulators: Renderer.summed_image / summed_sample_weights / summed_sample_counts /
 *    unidirectional_image_buffer (src/renderer.py:41-45).  Any pointer may be NULL. -- */
int cl2_read_accumulators(cl2_renderer* r, float* summed_image /*H*W*3*/, float* summed_sample_weights /*H*W*/,
                          int32_t* summed_sample_counts /*H*W*/, float* unidirectional /*H*W*3*/, size_t n_pixels);
int cl2_reset_accumulators(cl2_renderer* r);
/* -- output stage on the device: the tone map of src/camera.py:73-82 applied to one of the three pictures of
 *    src/renderer.py:293-316, straight from the accumulators (6 MB instead of 66 MB leave the device per 1080p frame;
 *    the host does no per-pixel work).  which: 0 `image` (summed_image / summed_sample_weights), 1 `unweighted_image`,
 *    2 `unidirectional_image`; all scrubbed with nan_to_num(neginf=0, posinf=0) and computed in the dtypes numpy's
 *    promotion gives the reference (csrc/tonemap.hpp).
 *      cl2_tone_log_sum   sum over the pixels of log(0.1 + luma) in float64 (the reference: np.sum(log_tone_sums));
 *                         the caller forms Lw = exp(sum / (H*W)) -- with numpy's exp if it wants numpy's last bit
 *      cl2_tone_map       (255 * result / (result + white_point^2)).astype(uint8), result = image * exposure / Lw, as
 *                         H*W*3 bytes, b, g, r per pixel
 *    Deterministic (fixed reduction tree).  The float64 sum is added in another order than numpy's pairwise sum, so Lw
 *    can differ from the host path's in its last bits; a byte of the picture changes only where 255*x/(x+w) lies within
 *    ~1e-13 of an integer.  `Renderer.image` keeps the host path (byte-exact against the reference's fixture);
 *    `Renderer.tone_mapped()` and movie.py use this one. -- */
int cl2_tone_log_sum(cl2_renderer* r, int which, double* sum_out);
int cl2_tone_map(cl2_renderer* r, int which, double exposure, double white_point, double log_average /* Lw */,
                 uint8_t* out_bgr, size_t n_bytes /* 3*H*W */);
/* packed planar form [8][H*W] = image b,g,r | weights | unidirectional b,g,r | counts(float):
 * the message of the multi-GPU sum-reduce, as host arrays (checkpointing, CPU-side tests). */
int cl2_read_accumulators_packed(cl2_renderer* r, float* host_dst, size_t n_floats);
int cl2_write_accumulators_packed(cl2_renderer* r, const float* host_src, size_t n_floats);

/* -- multi-GPU sample split (the reference is single-device, src/renderer.py:281-291).
 *    Samples are i.i.d. and the accumulators are pure sums (src/renderer.py:269-273): every rank
 *    renders its own samples of the replicated scene with its own seed buffer, then ONE in-place RCCL
 *    all-reduce (sum, float32) of the packed accumulators [8][H*W] combines them (66 MB at 1080p, over
 *    xGMI).  One communicator per handle, one handle per GPU, one process (or thread) per handle.
 *    librccl is loaded (dlopen) by the first of these calls; a failure is CL2_E_COMM, never fatal.
 *
 *      rank 0:  cl2_comm_get_unique_id(id)  -> hand the cl2_comm_unique_id_bytes() bytes to every rank
 *               (file, socket, launcher: the caller's business; clive2_amd/distributed.py uses a file)
 *      all   :  cl2_comm_init_rank(r, nranks, rank, id)      (collective: returns when all have called)
 *               ... cl2_run_samples(r, n_rank) ...
 *               cl2_reduce_accumulators(r)                  (collective; every rank then holds the sums)
 *               cl2_comm_destroy(r)                         (also done by cl2_destroy)
 *    failure:   cl2_comm_abort(r) on the rank that failed; the others get CL2_E_COMM from the collective -- */
int cl2_device_count(void);                          /* HIP devices visible to this process (0 on error) */
int cl2_synchronize(cl2_renderer* r);                /* drains the handle's streams, then the device */
int cl2_comm_unique_id_bytes(void);
int cl2_comm_get_unique_id(void* out_id, size_t n_bytes);           /* error text: cl2_last_error(NULL) */
int cl2_comm_init_rank(cl2_renderer* r, int nranks, int rank, const void* unique_id, size_t n_bytes);
int cl2_reduce_accumulators(cl2_renderer* r);
/* n <= 16 host doubles, in place, op 0 = sum, 1 = max over the ranks: barrier, max-over-ranks clock,
 * whole-job ray tally, error-flag agreement before the collective */
int cl2_comm_allreduce_f64(cl2_renderer* r, double* values, int n, int op);
int cl2_comm_destroy(cl2_renderer* r);
/* No collective blocks for ever: the wait for an enqueued all-reduce polls the stream and ncclCommGetAsyncError under
 * a deadline (CLIVE2_COMM_TIMEOUT_S seconds, default 300); on a timeout or an asynchronous error the communicator is
 * torn down with ncclCommAbort and the call returns CL2_E_COMM (the handle stays usable for local work, its
 * communicator is gone).  A rank that fails locally calls cl2_comm_abort before it exits, so that its peers do not
 * have to wait for their deadline; cl2_comm_destroy / cl2_destroy abort instead of destroying when the handle is in
 * a failed state. */
int cl2_comm_abort(cl2_renderer* r);

/* What the communicator reports about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and which GPU this
 * handle sits on (hipDeviceGetPCIBusId): the evidence that an N-rank job really spanned N devices (bench.py gathers the
 * addresses of all ranks into its JSON line).  Without a communicator nranks is 0 and the device fields are still filled. */
typedef struct {
    int32_t nranks, rank;           /* from the communicator; 0 / 0 without one */
    int32_t comm_device;            /* the device ordinal RCCL bound the communicator to */
    int32_t device_ordinal;         /* the ordinal this handle was created on */
    int64_t pci_address;            /* domain << 16 | bus << 8 | device << 3 | function */
    char pci_bus_id[32];            /* "0000:c1:00.0" */
} cl2_comm_info_t;
int cl2_comm_info(cl2_renderer* r, cl2_comm_info_t* out);

/* How the library organises the launches for the uploaded scene (all organisations give identical results;
 * this is what the automatic choices of cl2_set_traversal_mode / _levels_per_launch / _pipelining came to). */
typedef struct {
    int32_t tree_in_lds;            /* whole tree + all triangles staged in LDS by every workgroup (<= 512 records, <= 512 triangles) */
    int32_t persistent_subpaths;    /* subpath levels run as persistent traversal launches (+ one bounce launch per level) */
    int32_t persistent_connections; /* connection rays run as one persistent traversal launch */
    int32_t two_tris_per_step;      /* persistent walk tests two triangles per step (trees up to 16 MB) */
    int32_t n_records;              /* node records (>= reference boxes: leaves above 16 triangles are split) */
    int32_t n_lds_records;          /* records in the LDS window */
    int32_t n_top_renumbered;       /* boxes of the top levels numbered first so that the window holds them (0 = plain visit order) */
    int32_t lds_triangles;
    int32_t levels_per_launch;      /* effective subpath levels per launch */
    int32_t paths_share;            /* tuned share of the wave slots for the subpath stage, eighths (0 = not tuned yet, 9 = serial order won) */
    int32_t pipeline_stages;        /* effective setting of cl2_set_pipelining */
    int32_t wide_connections;       /* connection rays walk the exact 4-wide collapse of the tree (csrc/bvh_wide.hpp) */
    int32_t wide_nodes;             /* nodes of that collapse (0: not available: hand-made boxes that do not nest, leaves above 16 triangles) */
    int32_t pruned_records;         /* records of the pruned table of an LDS-resident tree (inner boxes whose test costs more than it saves are
                                       dropped: exact for rays with finite 1/d); 0: none */
    int64_t tree_bytes;             /* 32 B per record + 48 B per intersection triangle */
    int32_t sample_streams;         /* cl2_set_sample_streams */
    int32_t reserved;
} cl2_organisation;
int cl2_query_organisation(cl2_renderer* r, cl2_organisation* out);

/* -- counters / profiling -- */
int cl2_set_profiling(cl2_renderer* r, int level);  /* HIP-event timers: 0 off, 1 the traversal launches only (connection rays, subpath rays), 2 every stage */
/* mode 1: node / triangle test tallies of the REFERENCE's walk (src/trace.metal:144-176; the traversal kernels then run the
 * binary stackless walk, whose tests are the reference's one for one) -> cl2_counters.box_tests / tri_tests / counted_rays.
 * mode 2: what the exact 4-wide walk ITSELF fetches -> cl2_read_walk_tallies (the launches are the ones that run uncounted,
 * except that single run_sample() calls take the per-level organisation).  0: off. */
int cl2_set_counting(cl2_renderer* r, int mode);
int cl2_read_counters(cl2_renderer* r, cl2_counters* out);
typedef struct cl2_walk_tally {
    uint64_t rays, wide_visits, tri_records, stack_spills, binary_records;
} cl2_walk_tally;
typedef struct cl2_walk_tallies { cl2_walk_tally subpath, connection; } cl2_walk_tallies;
int cl2_read_walk_tallies(cl2_renderer* r, cl2_walk_tallies* out);
/* Launch-organisation switches for experiments and tests.  None of them changes a result.
 *   bit 3       accepted and ignored (rounds 2-3: the per-level subpath launches took the 4-wide walk in the serial order too; they
 *               always do since round 4)
 *   bits 4-6    7 = the second implementation of the resolve kernel, one wave per camera vertex (only in the test variant
 *               of the library); other values are refused
 *   bit 7       walk the full record table of an LDS-resident tree instead of the pruned one
 *   bits 8-10   eighths of the wave slots given to the subpath stage while the sample pipeline runs (0 = tuned)
 *   bit 11      walk a pruned table that is a plain list of leaves per lane instead of wave-uniformly (csrc/bvh_traverse.hpp,
 *               closest_hit_flat)
 *   bit 12      invert the one/two-triangles-per-step choice of the persistent walk
 *   bit 13      4-wide walk WITHOUT the speculative expansion of the stack top (round 5: a lane that is testing triangles expands
 *               the wide node on top of its stack in the same pass, csrc/bvh_wide.hpp): the pass as round 4 had it, for A/B runs and tests
 *   bit 14      4-wide walk with the 48-byte triangle records of the other walks (six 16-byte loads per pair) instead of its own
 *               36-byte ones, a pair fetched as one run of 72 bytes (five loads; round 6, csrc/bvh_wide.hpp: PACK), for A/B runs and tests
 *   bits 16-19  accepted and ignored (round 3: stack entries per lane in LDS of the 4-wide walk; a compile-time 8 since round 4)
 *   bits 20-23  4-wide walk: LDS window of the top of the wide tree in units of 32 nodes (0 = by tree size: 32 nodes, 64 when
 *               the tree streams from memory; 15 = no window)
 * Any other bit is refused (CL2_E_INVALID).  Bits 0-2 exist ONLY in the test variant of the library
 * (libclive2_amd_test.so, -DCL2_TEST_VARIANT), where they switch parts of the resolve stage off for timing
 * dissections -- bit 0 the t = 1 splat atomics, bit 1 / bit 2 the t >= 2 / t == 1 strategy pairs -- and make the
 * render INVALID; the shipped library refuses them. */
#define CL2_DEBUG_KNOWN_BITS 0x00FF7FFF
int cl2_set_debug_flags(cl2_renderer* r, int flags);
/* Reproducible light image, off by default.  The reference's light-image chain (sort by target pixel, per-pixel gather:
 * src/renderer.py:97-111, :213-250, src/trace.metal:872-964) is deterministic; the float atomics that replace it add a pixel's
 * contributions in hardware order, so two renders agree to a few ulp only.  on = 1: k_connect_resolve writes the reference's
 * records (slot id + s * total_pixels), one radix sort orders them by (target pixel, s, source pixel) and each target's run is
 * summed front to back -- two renders of the same scene and seeds give identical bytes in all four accumulators.
 * Memory: 32 B x 6 x B of records, keys and slot ids (B = sample streams x W x H entries; the sort runs over 6 x B keys) plus
 * rocPRIM's temporary storage -- 0.4 GB at 1920 x 1080 with one stream, 12.7 GB at 3840 x 2160 with 8.  Refused together with
 * debug bits 4-6 = 7 (the test variant's cross-check resolve kernel writes no records). */
int cl2_set_reproducible(cl2_renderer* r, int on);
int cl2_get_reproducible(const cl2_renderer* r);
/* Child order of the 4-wide walks (ABI 5).  order = 0 (default): the reference's fixed order -- a box's second child is popped first,
 * whatever the ray (src/trace.metal:157-160) -- which is what makes every output byte-comparable with the reference's.  order = 1:
 * the passing children of a node are taken NEAREST FIRST (by slab entry distance).  A closest-hit query then prunes what lies behind
 * its first hit: fewer node visits and triangle tests per ray.  NOT the reference's result by construction -- the reference's hit
 * depends on its visit order where a hit lies a few ulp in front of its own leaf box's entry distance (the leaf is entered or not
 * depending on what was found before it, trace.metal:152); two triangles hit at exactly the same t (first visited wins,
 * trace.metal:170) ARE settled as the reference settles them (a table of each triangle's position in its visit order) -- so: opt-in,
 * never the default, never the parity path; bench.py's headline and every parity test run with 0, and tests/test_gpu_round6.py /
 * test_gpu_fullsize.py count the rays whose hit differs (2 to 5 in 3.4e8) and check that each is such a hit.  Applies to
 * scenes whose tree is read through the caches (the 4-wide walk; an LDS-resident tree such as the Cornell box renders the same
 * either way).  No reference counterpart (src/trace.metal:144-176 has one order). */
int cl2_set_traversal_order(cl2_renderer* r, int order);
int cl2_get_traversal_order(const cl2_renderer* r);
/* Whole-subpath launch (traversal mode 4): lanes that must have gathered with a known closest hit before a wave runs
 * its bounce phase, and the steps the first of them waits at most.  0 = default (32 lanes, 48 steps).  Same results. */
int cl2_set_subpath_gather(cl2_renderer* r, int lanes, int wait_steps);
int cl2_reset_counters(cl2_renderer* r);

/* Exactness self-test: the kernels replace `1.0f/a` and `x/PI` by cheaper sequences that are proven
 * (exhaustively, over all 2^32 binary32 inputs) to return the correctly rounded IEEE result; this call
 * re-runs that proof on the device and returns the number of inputs that disagree (must be 0, 0). */
int cl2_selftest_exact_math(cl2_renderer* r, uint64_t* rcp_mismatches, uint64_t* divpi_mismatches);

/* Elementwise probe of the device's deterministic elementary functions: which = 0 sin, 1 cos, 2 acos,
 * 3 atan, 4 exp, 5 asin (csrc/detmath.hpp), 6 rcp_exact, 7 div_pi (csrc/vecmath.hpp).  Host arrays. */
int cl2_probe_math(cl2_renderer* r, int which, const float* in, size_t n, float* out);
/* Elementwise probe of the bounce routines (src/trace.metal:226-233, :254-264, :334-379).  in: 12 floats per
 * item {wi.xyz, n.xyz, rx, ry, ni, no, alpha, kind}, kind 0 diffuse / 1 reflect / 2 transmit / 3 GGX_sample
 * only; out: 8 floats {wo.xyz, f, c_p, l_p, fresnel(wi,m), m.x} with m = GGX_sample(n, rx, ry, alpha). */
int cl2_probe_bounce(cl2_renderer* r, int from_camera, const float* in, size_t n, float* out);

/* -- debug exports in the reference's AoS layouts (stage-level parity) -- */
int cl2_export_rays(cl2_renderer* r, int which, void* out_rays, size_t n_records);        /* Ray[batch]  */
int cl2_export_paths(cl2_renderer* r, int which, void* out_paths, size_t n_records);      /* Path[batch] */
int cl2_export_aggregators(cl2_renderer* r, void* out, size_t n_records);                 /* 128-B stride */
/* per-sample images: finalized_samples (float4), out_light_image rgb + summed light weight (float4),
 * sample_weights (K6 value only), out_camera_image (float4).  Any pointer may be NULL. */
int cl2_export_sample_images(cl2_renderer* r, float* finalized4, float* light4, float* sample_weights,
                             float* unidirectional4, size_t n_pixels);
/* the inverse of cl2_export_sample_images: per-sample images from host arrays (float4 / float per pixel),
 * so that the accumulation stage (process_images, src/renderer.py:253-278) can be checked on its own
 * against arrays produced by the reference's numpy code.  Any pointer may be NULL. */
int cl2_import_sample_images(cl2_renderer* r, const float* finalized4, const float* light4, const float* sample_weights,
                             const float* unidirectional4, size_t n_pixels);
/* closest-hit probe: n rays as Ray records -> (triangle, t, u, v) per ray; exercises the traversal
 * kernel alone (src/trace.metal:144-176). */
int cl2_probe_traverse(cl2_renderer* r, const void* rays, size_t n_rays, int32_t* best_i, float* best_t,
                       float* u, float* v);

#ifdef __cplusplus
}
#endif
#endif
