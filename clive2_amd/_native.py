"""ctypes binding of libclive2_amd.so (C ABI: include/clive2_amd.h).

The HIP library is the only implementation of the hot path: if it is missing or cannot be
loaded this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import glob

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libclive2_amd.so")
# the test variant additionally carries the second implementation of the resolve stage
# (tests/connect_resolve_wide.hpp, -DCL2_TEST_VARIANT): a cross-check for tests, not shipped code
TEST_LIB_PATH = os.path.join(_PKG, "libclive2_amd_test.so")
_MAIN_SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("renderer_api.hip", "bvh_builder_gpu.hip", "det_splat.hip")]
_HEADER = os.path.join(os.path.dirname(_PKG), "include", "clive2_amd.h")


_TEST_ONLY_SOURCES = [os.path.join(os.path.dirname(_PKG), "tests", "connect_resolve_wide.hpp")]


def _sources(variant=None):
    """Everything the translation unit includes: every file under csrc/ plus the public header (the test variant also
    includes the cross-check resolve kernel that lives under tests/)."""
    own = sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG, "csrc", "*.hpp"))) + [_HEADER]
    return own + (_TEST_ONLY_SOURCES if variant == "test" else [])

# -ffp-contract=off / no fast-math: every float op of the kernels rounds once, in source order.
# -fno-slp-vectorize: packed fp32 VALU ops (v_pk_mul/add_f32) issue at half the rate of scalar ones on
# gfx950 (tools/valu_rate.hip: 4.4 vs 2.3 cycles per wave-instruction), so SLP packing only adds the
# register shuffles that feed them.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
               "-fno-slp-vectorize", "-fPIC", "-shared"]


class RendererError(RuntimeError):
    """Single exception type of the binding (stands in for `metalcompute.error`, render.py:39)."""


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "conn_rays", "box_tests", "tri_tests", "counted_rays", "samples")] + \
               [(n, C.c_double) for n in ("ms_generate", "ms_traverse_paths", "ms_bounce", "ms_connect_setup",
                                          "ms_traverse_conn", "ms_connect_resolve", "ms_finalize", "ms_accumulate")] + \
               [(n, C.c_uint64) for n in ("launches_traverse_paths", "launches_traverse_conn",
                                          "rays_traverse_paths", "rays_traverse_conn")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Organisation(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("tree_in_lds", "persistent_subpaths", "persistent_connections", "two_tris_per_step",
                                         "n_records", "n_lds_records", "n_top_renumbered", "lds_triangles",
                                         "levels_per_launch", "paths_share", "pipeline_stages", "wide_connections", "wide_nodes", "pruned_records")] + \
               [("tree_bytes", C.c_int64), ("sample_streams", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class WalkTally(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "wide_visits", "tri_records", "stack_spills", "binary_records")]


class WalkTallies(C.Structure):
    _fields_ = [("subpath", WalkTally), ("connection", WalkTally)]


class CommInfo(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("comm_device", C.c_int32), ("device_ordinal", C.c_int32),
                ("pci_address", C.c_int64), ("pci_bus_id", C.c_char * 32)]


EXPORTS = [
    "cl2_create", "cl2_destroy", "cl2_last_error", "cl2_abi_version", "cl2_build_bvh", "cl2_build_bvh_gpu", "cl2_set_create_error", "cl2_upload_scene", "cl2_set_seeds",
    "cl2_get_seeds", "cl2_make_light_rays", "cl2_make_camera_rays", "cl2_trace_light_rays",
    "cl2_trace_camera_rays", "cl2_join_paths", "cl2_finalize_samples", "cl2_gather_light_image",
    "cl2_process_images", "cl2_run_samples", "cl2_set_levels_per_launch", "cl2_set_traversal_mode", "cl2_set_pipelining", "cl2_read_accumulators", "cl2_reset_accumulators",
    "cl2_read_accumulators_packed", "cl2_write_accumulators_packed", "cl2_device_count", "cl2_synchronize",
    "cl2_comm_unique_id_bytes", "cl2_comm_get_unique_id", "cl2_comm_init_rank", "cl2_reduce_accumulators",
    "cl2_comm_allreduce_f64", "cl2_comm_destroy", "cl2_import_sample_images", "cl2_query_organisation", "cl2_set_profiling", "cl2_set_counting", "cl2_set_debug_flags", "cl2_read_counters",
    "cl2_reset_counters", "cl2_selftest_exact_math", "cl2_export_rays", "cl2_export_paths", "cl2_export_aggregators",
    "cl2_export_sample_images", "cl2_probe_traverse", "cl2_probe_math", "cl2_probe_bounce",
    "cl2_tune", "cl2_set_subpath_gather", "cl2_comm_abort", "cl2_tone_log_sum", "cl2_tone_map",
    "cl2_set_sample_streams", "cl2_get_sample_streams", "cl2_set_export_stream", "cl2_comm_info",
    "cl2_read_walk_tallies", "cl2_set_reproducible", "cl2_get_reproducible", "cl2_set_traversal_order", "cl2_get_traversal_order",
]


def _path(variant):
    if variant not in (None, "test"):
        raise ValueError("library variant must be None or 'test'")
    return TEST_LIB_PATH if variant == "test" else LIB_PATH


def needs_build(variant=None):
    path = _path(variant)
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(s) > t for s in _sources(variant))


def build(force=False, verbose=False, variant=None):
    """Cross-compile the HIP library for gfx950 with hipcc (works without a GPU)."""
    path = _path(variant)
    if not force and not needs_build(variant):
        return path
    cmd = ["hipcc"] + HIPCC_FLAGS + (["-DCL2_TEST_VARIANT"] if variant == "test" else []) + _MAIN_SOURCES + ["-o", path]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RendererError("hipcc failed:\n" + res.stdout)
    if verbose:
        print(" ".join(cmd))
    return path


_libs = {}


def lib(variant=None):
    """Load the library (never builds implicitly: a box without the file has no render path).  It runs on
    the ROCm runtime it was linked against (/opt/rocm); nothing of torch is loaded or needed."""
    if variant not in _libs:
        path = _path(variant)
        if not os.path.exists(path):
            raise RendererError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
        # The ROCm runtime reads this when it initialises (first HIP call): on hosts whose driver only supports dmabuf IPC,
        # RCCL between processes fails with `hipIpcGetMemHandle: invalid argument` without it.  An explicit setting wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        L = C.CDLL(path)
        L.cl2_last_error.restype = C.c_char_p
        L.cl2_last_error.argtypes = [C.c_void_p]
        L.cl2_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cl2_destroy.argtypes = [C.c_void_p]
        L.cl2_destroy.restype = None
        L.cl2_comm_get_unique_id.argtypes = [C.c_void_p, C.c_size_t]
        L.cl2_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.cl2_comm_allreduce_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
        L.cl2_tune.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.cl2_set_subpath_gather.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cl2_set_sample_streams.argtypes = [C.c_void_p, C.c_int]
        L.cl2_get_sample_streams.argtypes = [C.c_void_p]
        L.cl2_set_traversal_order.argtypes = [C.c_void_p, C.c_int]
        L.cl2_get_traversal_order.argtypes = [C.c_void_p]
        L.cl2_set_export_stream.argtypes = [C.c_void_p, C.c_int]
        L.cl2_comm_info.argtypes = [C.c_void_p, C.POINTER(CommInfo)]
        L.cl2_tone_log_sum.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.cl2_tone_map.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_size_t]
        for name in ("cl2_reduce_accumulators", "cl2_comm_destroy", "cl2_comm_abort", "cl2_synchronize"):
            getattr(L, name).argtypes = [C.c_void_p]
        _libs[variant] = L
    return _libs[variant]


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
