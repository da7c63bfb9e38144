"""`Renderer`: the reference's render driver (src/renderer.py:16-352) on top of the HIP library.

Same constructor, stage-method names, `samples` counter, image properties and accumulator
attributes as the reference class, so callers written against it (`render.py:26-37`,
`movie.py:39-44`) work unchanged.  Differences, all deliberate:

* device state and accumulators live in the library (SoA in HBM); the accumulator attributes
  are read back on access instead of being updated on the host every sample;
* the RNG buffer is explicit: `Renderer(scene, seeds=...)` / `set_seeds()` (the reference seeds
  from unseeded `np.random.randint`, renderer.py:86-87, and is therefore not reproducible);
* `run_samples(n)` runs n iterations of the pipeline without returning to the host;
* `gather_light_image` is a no-op synchronisation point: the t=1 light-image splat happens in
  `join_paths` with float atomics (the reference's 300-launch bitonic sort + host bincount,
  renderer.py:212-250, has no counterpart);
* multi-GPU sample splitting: `comm_init(rank, world, id)` + `reduce_accumulators()` sum the
  accumulators of all ranks with one in-place RCCL all-reduce inside the library (no torch).
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import RendererError, Counters, Organisation, ptr
from . import struct_types as st
from .camera import tone_map
from .constants import timed, MAX_PATH_LENGTH  # noqa: F401

LIGHT, CAMERA = 0, 1


def make_seeds(batch_size, seed=20240928, rank=0):
    """Seed buffer of SURVEY.md §8(d): `RandomState(seed+rank).randint(0, 2**32, (B,2), uint32)`
    -- the call shape of renderer.py:86-87 -- with zeros (xorshift's fixed point) replaced by 1."""
    s = np.random.RandomState(seed + rank).randint(0, 2 ** 32, size=(batch_size, 2), dtype=np.uint32)
    s[s == 0] = 1
    return s


def stream_seeds(batch_size, streams, seed=20240928, first_rank=0):
    """Seed buffers of `streams` sample streams: stream k gets `make_seeds(batch, seed, first_rank + k)`, the buffer rank
    first_rank + k of a sample split would use.  Shape (batch, 2) for one stream, else (streams, batch, 2)."""
    if streams == 1:
        return make_seeds(batch_size, seed, first_rank)
    return np.stack([make_seeds(batch_size, seed, first_rank + k) for k in range(streams)])


def next_power_of_two(n):
    return 1 << (n - 1).bit_length() if n > 0 else 1


class Renderer:
    def __init__(self, scene, kernel_path=None, seeds=None, device=0, variant=None, streams=1):
        # kernel_path is accepted for signature compatibility (the reference JIT-compiles
        # trace.metal from it, renderer.py:27-29); the HIP kernels are precompiled.
        # variant="test" loads libclive2_amd_test.so (carries the cross-check resolve kernel).
        # streams=K: K independent samples of the frame per pass, one seed buffer each (what K reference Renderers --
        # the ranks of a sample split -- would render); seeds then has shape (K, batch, 2).  Default 1 = the reference;
        # "auto" = auto_streams() (with seeds=None: the default seed buffers of that many streams).
        self._h = C.c_void_p()
        self._L = _native.lib(variant)
        self.scene = scene
        self.device = device
        self.pixel_width, self.pixel_height = scene.pixel_width, scene.pixel_height
        self.batch_size = scene.pixel_width * scene.pixel_height
        rc = self._L.cl2_create(int(device), int(scene.pixel_width), int(scene.pixel_height), C.byref(self._h))
        if rc != 0:
            msg = self._L.cl2_last_error(None)
            self._h = C.c_void_p()
            raise RendererError(f"cl2_create failed ({rc}): {msg.decode() if msg else ''}")
        self.samples = 0
        self.streams = 1
        self.upload_scene(scene)
        if streams == "auto":
            streams = self.auto_streams()
        if int(streams) != 1:
            self.set_sample_streams(streams)
        self.set_seeds(stream_seeds(self.batch_size, self.streams) if seeds is None else seeds)

    # ---- plumbing ----
    _FATAL = (-2, -5)                    # CL2_E_HIP, CL2_E_COMM (include/clive2_amd.h): the device or the communicator is gone

    def _check(self, rc, what):
        if rc != 0:
            # close() aborts the communicator of a FAILED handle instead of waiting for peers.  A refused argument or a call out
            # of sequence (CL2_E_INVALID, CL2_E_STATE, CL2_E_NOMEM) leaves device and communicator healthy: a caller that catches
            # it and carries on must not tear the communicator down under peers that are fine.
            if rc in self._FATAL:
                self._failed = True
            msg = self._L.cl2_last_error(self._h)
            raise RendererError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def upload_scene(self, scene):
        if hasattr(scene, "validate"):
            scene.validate()
        def rec(a, dt):
            a = np.ascontiguousarray(a)
            if a.dtype.itemsize != dt.itemsize:
                raise RendererError(f"scene array has itemsize {a.dtype.itemsize}, expected {dt.itemsize}")
            return a.reshape(-1)
        boxes, tris = rec(scene.boxes, st.Box), rec(scene.triangles, st.Triangle)
        mats, cam = rec(scene.materials, st.Material), rec(scene.camera, st.Camera)
        ltris = rec(scene.light_triangles, st.Triangle)
        areas = np.ascontiguousarray(scene.light_surface_areas, dtype=np.float32).reshape(-1)
        lidx = np.ascontiguousarray(scene.light_triangle_indices, dtype=np.int32).reshape(-1)
        n_light = int(np.asarray(scene.light_counts).reshape(-1)[0])
        if not (len(ltris) == len(areas) == len(lidx) == n_light):
            raise RendererError("light arrays disagree in length")
        self._check(self._L.cl2_upload_scene(self._h, ptr(boxes), len(boxes), ptr(tris), len(tris), ptr(mats), len(mats),
                                             ptr(cam), ptr(ltris), ptr(areas), ptr(lidx), n_light), "cl2_upload_scene")

    def set_sample_streams(self, streams):
        """K independent samples of the frame per pass (cl2_set_sample_streams): frees and re-allocates the per-pixel device
        state, so the seeds must be set again -- `set_seeds` of shape (K, batch, 2), stream k = the buffer Renderer k of a
        sample split would own.  Accumulators, scene and counters are kept."""
        self._check(self._L.cl2_set_sample_streams(self._h, int(streams)), "cl2_set_sample_streams")
        self.streams = int(streams)

    def auto_streams(self):
        """The stream count `streams="auto"` takes: 1 for a scene whose tree is LDS-resident (one thread per pixel already fills
        the machine: the Cornell box gains nothing), else as many as bring a launch to about 2^24 entries, at most 8 (measured at
        1080p: config 3 9.7 / 11.2 / 11.4 Grays/s at 1 / 4 / 8 streams; at 3840 x 2160 two streams add 3 %)."""
        if self.organisation()["tree_in_lds"]:
            return 1
        return int(max(1, min(8, round((1 << 24) / self.batch_size))))

    def set_export_stream(self, stream):
        """The sample stream that export_rays / export_paths / export_aggregators / {ex,im}port_sample_images address."""
        self._check(self._L.cl2_set_export_stream(self._h, int(stream)), "cl2_set_export_stream")

    def set_seeds(self, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        if s.size != 2 * self.batch_size * self.streams:
            raise RendererError(f"seed buffer needs {2 * self.batch_size * self.streams} uint32 words, got {s.size}")
        self._check(self._L.cl2_set_seeds(self._h, ptr(s), C.c_size_t(s.size)), "cl2_set_seeds")

    def get_random_buffer(self):
        """(batch, 2) uint32 as the reference's rand_buffer; (streams, batch, 2) with more than one sample stream."""
        shape = (self.batch_size, 2) if self.streams == 1 else (self.streams, self.batch_size, 2)
        s = np.empty(shape, dtype=np.uint32)
        self._check(self._L.cl2_get_seeds(self._h, ptr(s), C.c_size_t(s.size)), "cl2_get_seeds")
        return s

    # ---- the eight stages of run_sample (renderer.py:113-278) ----
    @timed
    def make_light_rays(self):
        self._check(self._L.cl2_make_light_rays(self._h), "make_light_rays")

    @timed
    def make_camera_rays(self):
        self._check(self._L.cl2_make_camera_rays(self._h), "make_camera_rays")

    @timed
    def trace_light_rays(self):
        self._check(self._L.cl2_trace_light_rays(self._h), "trace_light_rays")

    @timed
    def trace_camera_rays(self):
        self._check(self._L.cl2_trace_camera_rays(self._h), "trace_camera_rays")

    @timed
    def join_paths(self):
        self._check(self._L.cl2_join_paths(self._h), "join_paths")

    @timed
    def finalize_samples(self):
        self._check(self._L.cl2_finalize_samples(self._h), "finalize_samples")

    @timed
    def gather_light_image(self):
        self._check(self._L.cl2_gather_light_image(self._h), "gather_light_image")

    @timed
    def process_images(self):
        self._check(self._L.cl2_process_images(self._h), "process_images")

    @timed
    def run_sample(self):
        self.run_samples(1)

    def run_samples(self, n):
        """n passes of the pipeline = n x run_sample (n * streams samples of the frame with more than one sample stream)."""
        self._check(self._L.cl2_run_samples(self._h, int(n)), "run_samples")
        self.samples += int(n) * self.streams

    # ---- accumulators (renderer.py:41-45) ----
    def read_accumulators(self):
        H, W, B = self.pixel_height, self.pixel_width, self.batch_size
        img = np.empty((H, W, 3), np.float32)
        wts = np.empty((H, W, 1), np.float32)
        cnt = np.empty((H, W, 1), np.int32)
        uni = np.empty((H, W, 3), np.float32)
        self._check(self._L.cl2_read_accumulators(self._h, ptr(img), ptr(wts), ptr(cnt), ptr(uni), C.c_size_t(B)),
                    "cl2_read_accumulators")
        return img, wts, cnt, uni

    summed_image = property(lambda self: self.read_accumulators()[0])
    summed_sample_weights = property(lambda self: self.read_accumulators()[1])
    summed_sample_counts = property(lambda self: self.read_accumulators()[2])
    unidirectional_image_buffer = property(lambda self: self.read_accumulators()[3])

    def reset_accumulators(self):
        self._check(self._L.cl2_reset_accumulators(self._h), "cl2_reset_accumulators")
        self.samples = 0

    @property
    def radiance(self):
        """`summed_image / summed_sample_weights`, scrubbed -- the reference image before tone
        mapping (renderer.py:295-297); float32 (H,W,3), BGR."""
        img, wts, _, _ = self.read_accumulators()
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(img / wts, neginf=0, posinf=0)

    @property
    def unidirectional_radiance(self):
        _, _, cnt, uni = self.read_accumulators()
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(uni / cnt, neginf=0, posinf=0)

    @property
    def image(self):
        return tone_map(self.radiance, exposure=4.0)

    _PICTURES = {"image": 0, "unweighted_image": 1, "unidirectional_image": 2}

    def tone_mapped(self, which="image", exposure=4.0, white_point=1.0):
        """`tone_map` (camera.py:73-82) of one of the three pictures of renderer.py:293-316, computed ON THE DEVICE from
        the accumulators (cl2_tone_log_sum + cl2_tone_map, csrc/tonemap.hpp): uint8 (H,W,3), BGR.  Only the 3*W*H bytes
        of the picture cross PCIe (the host path reads the 32*W*H bytes of the accumulators and maps them with numpy).
        Same arithmetic and dtypes as the host path; the float64 log-luminance sum is added in another order, so a byte
        can differ from `.image` where 255*x/(x+w) lies within ~1e-13 of an integer."""
        w = self._PICTURES[which]
        # the device arithmetic restates what numpy >= 2 (NEP 50 promotion) makes of camera.py:73-82: the float32 picture times
        # the exposure stays float32 and is widened by the division by the float64 Lw.  numpy 1.x keeps `result` in float32
        # (value-based casting), and the host path then differs from this one by far more than one count.
        if int(np.__version__.split(".")[0]) < 2:
            raise RendererError("tone_mapped() reproduces the reference's tone map under numpy >= 2 (NEP 50) only; use .image")
        s = C.c_double(0.0)
        self._check(self._L.cl2_tone_log_sum(self._h, w, C.byref(s)), "cl2_tone_log_sum")
        log_avg = np.exp(np.float64(s.value) / (self.pixel_height * self.pixel_width))       # Lw, with numpy's exp
        out = np.empty((self.pixel_height, self.pixel_width, 3), np.uint8)
        self._check(self._L.cl2_tone_map(self._h, w, float(exposure), float(white_point), float(log_avg), ptr(out),
                                         C.c_size_t(out.size)), "cl2_tone_map")
        return out

    @property
    def unweighted_image(self):
        return tone_map(np.nan_to_num(self.read_accumulators()[0], neginf=0, posinf=0), exposure=4.0)

    @property
    def unidirectional_image(self):
        return tone_map(self.unidirectional_radiance, exposure=4.0)

    # ---- multi-GPU sample split: one sum-reduce of the packed accumulators ----
    def packed_accumulators(self):
        a = np.empty(8 * self.batch_size, np.float32)
        self._check(self._L.cl2_read_accumulators_packed(self._h, ptr(a), C.c_size_t(a.size)), "read_packed")
        return a

    def load_packed_accumulators(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        self._check(self._L.cl2_write_accumulators_packed(self._h, ptr(a), C.c_size_t(a.size)), "write_packed")

    def comm_init(self, rank, world_size, unique_id):
        """Join the RCCL communicator of the sample split (collective: returns when every rank has
        called it).  `unique_id`: the bytes rank 0 got from `comm_unique_id()` and handed to everybody
        (`clive2_amd.distributed.exchange_unique_id`)."""
        uid = bytes(unique_id)
        self._check(self._L.cl2_comm_init_rank(self._h, int(world_size), int(rank), uid, C.c_size_t(len(uid))), "cl2_comm_init_rank")
        self.rank, self.world_size = int(rank), int(world_size)

    def comm_unique_id(self):
        n = self._L.cl2_comm_unique_id_bytes()
        buf = C.create_string_buffer(n)
        rc = self._L.cl2_comm_get_unique_id(buf, C.c_size_t(n))
        if rc != 0:
            msg = self._L.cl2_last_error(None)
            raise RendererError(f"cl2_comm_get_unique_id failed ({rc}): {msg.decode() if msg else ''}")
        return buf.raw

    def reduce_accumulators(self):
        """ONE in-place all-reduce (sum) of the packed accumulators over the communicator's ranks
        (RCCL over xGMI); every rank then holds the sums.  Needs `comm_init`."""
        self._check(self._L.cl2_reduce_accumulators(self._h), "cl2_reduce_accumulators")

    def allreduce_host(self, values, op="sum"):
        """Up to 16 host floats summed / maximised over the ranks (also the barrier of the job)."""
        v = (C.c_double * len(values))(*[float(x) for x in values])
        self._check(self._L.cl2_comm_allreduce_f64(self._h, v, len(values), {"sum": 0, "max": 1}[op]), "cl2_comm_allreduce_f64")
        return list(v)

    def comm_info(self):
        """What the communicator says about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and the PCI address
        of this handle's GPU; `nranks` is 0 without a communicator."""
        info = _native.CommInfo()
        self._check(self._L.cl2_comm_info(self._h, C.byref(info)), "cl2_comm_info")
        return {"nranks": info.nranks, "rank": info.rank, "comm_device": info.comm_device, "device_ordinal": info.device_ordinal,
                "pci_address": info.pci_address, "pci_bus_id": info.pci_bus_id.decode()}

    def comm_destroy(self):
        self._check(self._L.cl2_comm_destroy(self._h), "cl2_comm_destroy")

    def comm_abort(self):
        """This rank failed: tear the communicator down without waiting for the others (ncclCommAbort)."""
        self._L.cl2_comm_abort(self._h)

    def synchronize(self):
        self._check(self._L.cl2_synchronize(self._h), "cl2_synchronize")

    # ---- counters / profiling ----
    def set_profiling(self, level=2):
        """HIP-event stage timers: 0 off, 1 the connection-ray traversal launch only, 2 (or True) every stage."""
        level = 2 if level is True else int(level)
        self._check(self._L.cl2_set_profiling(self._h, level), "set_profiling")

    def set_counting(self, on=True):
        """True / 1: the node and triangle test tallies of the reference's walk (counters()); 2: what the 4-wide walk itself
        fetches (walk_tallies()); False / 0: off."""
        self._check(self._L.cl2_set_counting(self._h, int(on)), "set_counting")

    def walk_tallies(self):
        """{'subpath': {...}, 'connection': {...}} of set_counting(2): rays, wide nodes visited, triangle records read, stack
        entries spilled to global memory, binary records (rays with a non-finite 1/d) since the last reset_counters()."""
        t = _native.WalkTallies()
        self._check(self._L.cl2_read_walk_tallies(self._h, C.byref(t)), "read_walk_tallies")
        return {k: {n: getattr(getattr(t, k), n) for n, _ in _native.WalkTally._fields_} for k in ("subpath", "connection")}

    def set_levels_per_launch(self, levels):
        """Bounces traced per launch (1..6, 0 = chosen from the survival seen in the first sample); a pure
        performance knob, results are identical."""
        self._check(self._L.cl2_set_levels_per_launch(self._h, int(levels)), "set_levels_per_launch")

    def selftest_exact_math(self):
        """(rcp mismatches, x/pi mismatches) over all 2^32 float inputs; both must be 0."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.cl2_selftest_exact_math(self._h, C.byref(a), C.byref(b)), "selftest_exact_math")
        return a.value, b.value

    def set_pipelining(self, stages=-1):
        """Sample pipeline inside run_samples: 0 serial, 1 subpaths of sample i+1 beside the connection
        phase of sample i, 2 three stages, -1 (default) chosen by frame size.  A pure performance knob,
        results are identical."""
        self._check(self._L.cl2_set_pipelining(self._h, int(stages)), "set_pipelining")

    def set_traversal_mode(self, mode):
        """0 auto, 1 fused one-ray-per-lane kernels, 2 persistent traversal with ray replacement (one launch per
        level), 3 fused subpaths + persistent connection rays, 4 whole subpaths in one persistent launch, 5 exact 4-wide
        walk for the connection rays."""
        self._check(self._L.cl2_set_traversal_mode(self._h, int(mode)), "set_traversal_mode")

    def set_debug_flags(self, flags):
        """Launch-organisation switches (include/clive2_amd.h); none changes a result.  The shipped library
        refuses bits 0-2 (they skip parts of the resolve stage and exist only in the test variant)."""
        self._check(self._L.cl2_set_debug_flags(self._h, int(flags)), "set_debug_flags")

    def set_traversal_order(self, order=1):
        """Child order of the 4-wide walks (cl2_set_traversal_order): 0 = the reference's fixed order (trace.metal:157-160; the
        default, bit-exact), 1 = nearest child first -- fewer node visits and triangle tests per ray; exact-t ties go to the triangle
        the reference meets first, but NOT bit-exact by construction (a hit that rounding puts in front of its own leaf box is found
        or not depending on the visit order: a few rays in 1e8).  Opt-in; no reference counterpart."""
        self._check(self._L.cl2_set_traversal_order(self._h, int(order)), "set_traversal_order")

    def traversal_order(self):
        return int(self._L.cl2_get_traversal_order(self._h))

    def set_reproducible(self, on=True):
        """Reproducible light image (cl2_set_reproducible): the t = 1 contributions sorted and summed in a fixed order instead of
        float atomics -- two renders of the same scene and seeds then agree byte for byte, as the reference's sort + gather
        chain does (renderer.py:97-111, :213-250).  Off by default (costs a radix sort per pass)."""
        self._check(self._L.cl2_set_reproducible(self._h, int(bool(on))), "set_reproducible")

    def set_subpath_gather(self, lanes=0, wait_steps=0):
        """Whole-subpath launch: lanes gathered / steps waited before a wave runs its bounce phase (0 = default)."""
        self._check(self._L.cl2_set_subpath_gather(self._h, int(lanes), int(wait_steps)), "set_subpath_gather")

    def tune(self):
        """Make the measured launch-organisation choices now (cl2_tune); returns the number of (real) samples it rendered."""
        n = C.c_int(0)
        self._check(self._L.cl2_tune(self._h, C.byref(n)), "cl2_tune")
        self.samples += n.value * self.streams
        return n.value

    def counters(self):
        c = Counters()
        self._check(self._L.cl2_read_counters(self._h, C.byref(c)), "read_counters")
        return c.as_dict()

    def organisation(self):
        """What the automatic launch-organisation choices came to for this scene (dict of cl2_organisation)."""
        o = Organisation()
        self._check(self._L.cl2_query_organisation(self._h, C.byref(o)), "query_organisation")
        return o.as_dict()

    def reset_counters(self):
        self._check(self._L.cl2_reset_counters(self._h), "reset_counters")

    # ---- debug exports in the reference's AoS layouts ----
    def export_rays(self, which):
        out = np.zeros(self.batch_size, dtype=st.Ray)
        self._check(self._L.cl2_export_rays(self._h, int(which), ptr(out), C.c_size_t(len(out))), "export_rays")
        return out

    def export_paths(self, which):
        out = np.zeros(self.batch_size, dtype=st.Path)
        self._check(self._L.cl2_export_paths(self._h, int(which), ptr(out), C.c_size_t(len(out))), "export_paths")
        return out

    def export_aggregators(self):
        out = np.zeros(self.batch_size, dtype=st.WeightAggregator)
        self._check(self._L.cl2_export_aggregators(self._h, ptr(out), C.c_size_t(len(out))), "export_aggregators")
        return out

    def export_sample_images(self):
        B = self.batch_size
        fin, light, uni = (np.zeros((B, 4), np.float32) for _ in range(3))
        sw = np.zeros(B, np.float32)
        self._check(self._L.cl2_export_sample_images(self._h, ptr(fin), ptr(light), ptr(sw), ptr(uni), C.c_size_t(B)),
                    "export_sample_images")
        return dict(finalized=fin, light=light, sample_weights=sw, unidirectional=uni)

    def import_sample_images(self, finalized=None, light=None, sample_weights=None, unidirectional=None):
        """Per-sample images from host arrays ((B,4) float32 / (B,) float32): lets `process_images` be
        checked on its own against the reference's numpy code (tests/golden/renderer_glue.npz)."""
        def f4(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float32).reshape(self.batch_size, 4)
        fin, li, un = f4(finalized), f4(light), f4(unidirectional)
        sw = None if sample_weights is None else np.ascontiguousarray(sample_weights, dtype=np.float32).reshape(self.batch_size)
        self._check(self._L.cl2_import_sample_images(self._h, ptr(fin), ptr(li), ptr(sw), ptr(un), C.c_size_t(self.batch_size)),
                    "import_sample_images")

    def probe_traverse(self, rays):
        rays = np.ascontiguousarray(rays, dtype=st.Ray)
        n = len(rays)
        bi, bt = np.empty(n, np.int32), np.empty(n, np.float32)
        u, v = np.empty(n, np.float32), np.empty(n, np.float32)
        self._check(self._L.cl2_probe_traverse(self._h, ptr(rays), C.c_size_t(n), ptr(bi), ptr(bt), ptr(u), ptr(v)),
                    "probe_traverse")
        return bi, bt, u, v

    def probe_math(self, which, x):
        """Device detmath / exact-reciprocal functions on a float32 array (`which`: sin cos acos atan exp asin rcp div_pi)."""
        code = ["sin", "cos", "acos", "atan", "exp", "asin", "rcp", "div_pi"].index(which)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self._check(self._L.cl2_probe_math(self._h, code, ptr(x), C.c_size_t(x.size), ptr(out)), "probe_math")
        return out

    def probe_bounce(self, items, from_camera=True):
        """Device bounce routines on an (n,12) float32 array {wi, n, rx, ry, ni, no, alpha, kind} -> (n,8)."""
        items = np.ascontiguousarray(items, dtype=np.float32).reshape(-1, 12)
        out = np.empty((len(items), 8), np.float32)
        self._check(self._L.cl2_probe_bounce(self._h, int(bool(from_camera)), ptr(items), C.c_size_t(len(items)), ptr(out)),
                    "probe_bounce")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_failed", False):
                self._L.cl2_comm_abort(self._h)
            self._L.cl2_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
