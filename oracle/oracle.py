"""oracle.py -- TEST INFRASTRUCTURE.  ctypes front-end of the C oracle + the reference's host glue.

`OracleRenderer` restates `Renderer` of the reference (`src/renderer.py:16-316`) on the CPU:
the eight Metal kernels are the C functions of `bdpt_oracle.c`, every buffer is a numpy array
in the reference's AoS layout, and the two pieces of host arithmetic the reference itself does
in numpy -- `light_bins` (:97-111) and `process_images` (:253-278) -- are numpy here too.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import this.
Nothing here touches /root/reference at run time.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# Record dtypes: restated here so the oracle does not depend on the product package.
_F3 = (np.float32, (4,))
Ray = np.dtype([("origin", *_F3), ("direction", *_F3), ("inv_direction", *_F3), ("color", *_F3),
                ("normal", *_F3), ("material", "<i4"), ("triangle", "<i4"), ("c_importance", "<f4"),
                ("l_importance", "<f4"), ("tot_importance", "<f4"), ("hit_light", "<i4"),
                ("from_camera", "<i4"), ("hit_camera", "<i4"), ("pixel_idx", "<i4"), ("pad", "<i4", (3,))])
Path = np.dtype([("rays", Ray, (8,)), ("length", "<i4"), ("from_camera", "<i4"), ("pad", "<i4", (2,))])
WeightAggregator = np.dtype({"names": ["weights", "total_contribution", "contrib_weight_sum"],
                             "formats": [("<f4", (3, 3)), ("<f4", (4,)), "<f4"],
                             "offsets": [0, 48, 64], "itemsize": 128})
Counters = np.dtype([("rays", "<u8"), ("box_tests", "<u8"), ("tri_tests", "<u8")])

MAX_PATH_LENGTH = 8      # renderer.py:8


def build(force=False):
    """Compile liboracle.so / liboracle_libm.so with the recipe in oracle/Makefile."""
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle.so", "liboracle_libm.so")):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


_libs = {}


def lib(libm=False):
    key = "liboracle_libm.so" if libm else "liboracle.so"
    if key not in _libs:
        path = os.path.join(_HERE, key)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_xorshift.restype = C.c_float
        L.orc_fresnel.restype = C.c_float
        L.orc_ggx_d.restype = C.c_float
        assert L.orc_sizeof(0) == Ray.itemsize and L.orc_sizeof(1) == Path.itemsize
        assert L.orc_sizeof(6) == WeightAggregator.itemsize
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def next_power_of_two(n):
    return 1 << (n - 1).bit_length() if n > 0 else 1


def xorshift_floats(seed, count):
    """Known-answer helper: `count` successive xorshift_random outputs (trace.metal:87-93)."""
    s = C.c_uint32(seed)
    L = lib()
    out = []
    for _ in range(count):
        f = L.orc_xorshift(C.byref(s))
        out.append((s.value, np.float32(f)))
    return out


def det_math(which, x, libm=False):
    names = {"sin": 0, "cos": 1, "acos": 2, "atan": 3, "exp": 4, "asin": 5}
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib(libm).orc_math(names[which], C.c_int(x.size), _p(x), _p(out))
    return out


def traverse(rays, boxes, triangles):
    """Closest hit for each Ray record (trace.metal:144-176) -> (best_i, best_t, u, v, counters)."""
    rays = np.ascontiguousarray(rays)
    n = len(rays)
    bi = np.empty(n, np.int32)
    bt = np.empty(n, np.float32)
    u = np.empty(n, np.float32)
    v = np.empty(n, np.float32)
    cnt = np.zeros(1, Counters)
    lib().orc_traverse(C.c_int(n), _p(rays), _p(np.ascontiguousarray(boxes)), _p(np.ascontiguousarray(triangles)),
                       _p(bi), _p(bt), _p(u), _p(v), _p(cnt))
    return bi, bt, u, v, cnt[0]


def strategy_log(renderer):
    """Runs `renderer.join_paths()` with the strategy log on: returns a float32 array [B, 7, 7, 24], record [id, t, s] =
    {1.0 if the pair produced a weight, w, p_s, sum(p_values), g, color.xyz, s+t, p_values[0..12], pad} (bdpt_oracle.c:
    orc_set_strategy_log).  Tests only."""
    L = renderer.L
    stride = L.orc_strategy_log_stride()
    buf = np.zeros((renderer.batch_size, 7, 7, stride), np.float32)
    L.orc_set_strategy_log(_p(buf))
    try:
        renderer.join_paths()
    finally:
        L.orc_set_strategy_log(None)
    return buf


def stage_hashes(renderer):
    """SHA-256 of what one full sample leaves behind on a renderer (oracle or product: both expose the same arrays through
    the same names): both Path[] buffers, the filter aggregators, the RNG buffer.  The drift pin of tests/golden/."""
    import hashlib
    def h(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    return {"light_paths": h(renderer.out_light_paths), "camera_paths": h(renderer.out_camera_paths),
            "aggregator_total": h(renderer.weight_aggregators["total_contribution"]),
            "aggregator_weights": h(renderer.weight_aggregators["weights"]),
            "aggregator_weight_sum": h(renderer.weight_aggregators["contrib_weight_sum"]),
            "seeds": h(renderer.rand_buffer)}


def make_seeds(batch, seed=20240928, rank=0):
    """Seed buffer of SURVEY.md §8(d): same call shape as renderer.py:86-87, but seeded, and with
    zeros (the xorshift fixed point) replaced by 1."""
    s = np.random.RandomState(seed + rank).randint(0, 2 ** 32, size=(batch, 2), dtype=np.uint32)
    s[s == 0] = 1
    return s


class OracleRenderer:
    """CPU restatement of reference `Renderer` (renderer.py:16-316).  `scene` is any object with the
    reference's Scene attributes as numpy arrays (clive2_amd.scene.Scene qualifies)."""

    def __init__(self, scene, seeds=None, libm=False):
        self.L = lib(libm)
        self.scene = scene
        self.pixel_width, self.pixel_height = scene.pixel_width, scene.pixel_height
        B = self.batch_size = scene.pixel_width * scene.pixel_height
        res = (scene.pixel_height, scene.pixel_width)
        self.boxes = np.ascontiguousarray(scene.boxes)
        self.triangles = np.ascontiguousarray(scene.triangles)
        self.materials = np.ascontiguousarray(scene.materials)
        self.camera = np.ascontiguousarray(scene.camera).reshape(-1)
        self.light_triangles = np.ascontiguousarray(scene.light_triangles)
        self.light_surface_areas = np.ascontiguousarray(scene.light_surface_areas, dtype=np.float32)
        self.light_triangle_indices = np.ascontiguousarray(scene.light_triangle_indices, dtype=np.int32)
        self.light_counts = np.ascontiguousarray(scene.light_counts, dtype=np.int32).reshape(1)

        # host accumulators (renderer.py:41-45)
        self.summed_image = np.zeros((*res, 3), np.float32)
        self.summed_sample_counts = np.zeros((*res, 1), np.int32)
        self.summed_sample_weights = np.zeros((*res, 1), np.float32)
        self.unidirectional_image_buffer = np.zeros((*res, 3), np.float32)

        # "device" buffers (renderer.py:51-80)
        self.camera_ray_buffer = np.zeros(B, Ray)
        self.light_ray_buffer = np.zeros(B, Ray)
        self.indices_buffer = np.arange(B, dtype=np.uint32)              # assign_indices, :89-94
        self.summed_bins_buffer = np.arange(B + 1, dtype=np.uint32)
        self.rand_buffer = (make_seeds(B) if seeds is None else np.array(seeds, dtype=np.uint32, copy=True)).reshape(B, 2)
        self.out_camera_image = np.zeros((B, 4), np.float32)
        self.out_camera_paths = np.zeros(B, Path)
        self.out_camera_debug_image = np.zeros((B, 4), np.float32)
        self.out_samples = np.zeros((B, 4), np.float32)
        self.n_light = next_power_of_two(B * MAX_PATH_LENGTH)
        self.out_light_indices = np.zeros(self.n_light, np.int32)
        self.out_light_path_indices = np.zeros(self.n_light, np.int32)
        self.out_light_ray_indices = np.zeros(self.n_light, np.int32)
        self.out_light_weights = np.zeros(self.n_light, np.float32)
        self.out_light_shade = np.zeros(self.n_light, np.float32)
        self.weight_aggregators = np.zeros(B, WeightAggregator)
        self.finalized_samples = np.zeros((B, 4), np.float32)
        self.sample_counts = np.zeros(B, np.uint32)
        self.sample_weights = np.zeros(B, np.float32)
        self.out_light_image = np.zeros((B, 4), np.float32)
        self.out_light_paths = np.zeros(B, Path)
        self.out_light_debug_image = np.zeros((B, 4), np.float32)
        self.counters = np.zeros(1, Counters)
        self.samples = 0

    # -- stages, same names and order as renderer.py:113-278 --
    def make_light_rays(self):
        self.L.orc_generate_light_rays(C.c_int(self.batch_size), _p(self.light_triangles), _p(self.light_surface_areas),
                                       _p(self.light_triangle_indices), _p(self.materials), _p(self.rand_buffer),
                                       _p(self.light_ray_buffer), _p(self.light_counts))

    def make_camera_rays(self):
        self.L.orc_generate_camera_rays(C.c_int(self.batch_size), _p(self.camera), _p(self.rand_buffer),
                                        _p(self.indices_buffer), _p(self.camera_ray_buffer))

    def _trace(self, rays, out_image, out_paths, out_debug):
        self.L.orc_generate_paths(C.c_int(self.batch_size), _p(rays), _p(self.boxes), _p(self.triangles),
                                  _p(self.materials), _p(self.rand_buffer), _p(out_image), _p(out_paths),
                                  _p(out_debug), _p(self.counters))

    def trace_camera_rays(self):
        self._trace(self.camera_ray_buffer, self.out_camera_image, self.out_camera_paths, self.out_camera_debug_image)

    def trace_light_rays(self):
        self._trace(self.light_ray_buffer, self.out_light_image, self.out_light_paths, self.out_light_debug_image)

    def _light_arrays(self):
        return (_p(self.out_light_indices), _p(self.out_light_path_indices), _p(self.out_light_ray_indices),
                _p(self.out_light_weights), _p(self.out_light_shade))

    def join_paths(self):
        self.L.orc_reset_light_indices(C.c_int64(self.n_light), *self._light_arrays())
        self.L.orc_connect_paths(C.c_int(self.batch_size), _p(self.out_camera_paths), _p(self.out_light_paths),
                                 _p(self.triangles), _p(self.materials), _p(self.boxes), _p(self.camera),
                                 _p(self.weight_aggregators), _p(self.out_samples), *self._light_arrays(),
                                 _p(self.counters))

    def finalize_samples(self):
        self.L.orc_adaptive_finalize_samples(C.c_int(self.batch_size), _p(self.weight_aggregators), _p(self.camera),
                                             _p(self.finalized_samples), _p(self.sample_counts),
                                             _p(self.summed_bins_buffer), _p(self.sample_weights))

    def light_bins(self):
        """renderer.py:97-111, verbatim semantics."""
        idx = self.out_light_indices
        bins = np.bincount(idx[idx >= 0], minlength=self.pixel_height * self.pixel_width)
        summed = np.insert(np.cumsum(bins), 0, 0).astype(np.uint32)
        offset = np.sum(idx < 0).astype(np.uint32)
        return summed, offset

    def gather_light_image(self, stable=False):
        """renderer.py:212-250.  `stable=True` (tests of the product's reproducible light image): the five arrays are ordered
        by a STABLE sort on the pixel key instead of the reference's bitonic network, i.e. a pixel's run keeps slot order
        `id + s * total_pixels` = by (s, source pixel); bins and K8 (trace.metal:937-964) are unchanged."""
        if stable:
            order = np.argsort(self.out_light_indices, kind="stable")
            for a in (self.out_light_indices, self.out_light_path_indices, self.out_light_ray_indices,
                      self.out_light_weights, self.out_light_shade):
                a[:] = a[order]
        else:
            self.L.orc_light_sort_all(*self._light_arrays(), C.c_uint32(self.n_light))
        bins, offset = self.light_bins()
        bins = np.ascontiguousarray(bins.astype(np.int32))
        self.L.orc_light_image_gather(C.c_int(self.batch_size), _p(self.out_light_paths), _p(self.materials),
                                      _p(self.out_light_path_indices), _p(self.out_light_ray_indices), _p(bins),
                                      C.c_uint32(int(offset)), _p(self.out_light_weights), _p(self.out_light_shade),
                                      _p(self.out_light_image), _p(self.sample_weights))

    def process_images(self):
        """renderer.py:253-278."""
        H, W = self.pixel_height, self.pixel_width
        finalized = self.finalized_samples.reshape(H, W, 4)[:, :, :3]
        light = self.out_light_image.reshape(H, W, 4)[:, :, :3]
        image = light + finalized
        self.summed_image += np.nan_to_num(image, posinf=0, neginf=0)
        self.summed_sample_counts += self.sample_counts.view(np.int32).reshape(H, W, 1)
        self.summed_sample_weights += self.sample_weights.reshape(H, W, 1)
        uni = self.out_camera_image.reshape(H, W, 4)[:, :, :3]
        self.unidirectional_image_buffer += np.nan_to_num(uni, posinf=0, neginf=0)

    def run_sample(self, stable_light_sort=False):
        self.make_light_rays()
        self.make_camera_rays()
        self.trace_light_rays()
        self.trace_camera_rays()
        self.join_paths()
        self.finalize_samples()
        self.gather_light_image(stable=stable_light_sort)
        self.process_images()
        self.samples += 1

    # -- pre-tone-map images (renderer.py:293-316 without tone_map) --
    @property
    def radiance(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(self.summed_image / self.summed_sample_weights, neginf=0, posinf=0)

    @property
    def unidirectional_radiance(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(self.unidirectional_image_buffer / self.summed_sample_counts, neginf=0, posinf=0)

    @property
    def rays_traced(self):
        return int(self.counters["rays"][0])
