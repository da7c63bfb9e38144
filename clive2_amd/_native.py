"""ctypes binding of libclive2_amd.so (C ABI: include/clive2_amd.h).

The HIP library is the only implementation of the hot path: if it is missing or cannot be
loaded this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libclive2_amd.so")
_SOURCES = [os.path.join(_PKG, "csrc", n) for n in
            ("renderer_api.hip", "kernels.hpp", "connect_resolve.hpp", "bvh_traverse.hpp", "bvh_builder.hpp", "bsdf.hpp",
             "vecmath.hpp", "detmath.hpp")]
_HEADER = os.path.join(os.path.dirname(_PKG), "include", "clive2_amd.h")

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


EXPORTS = [
    "cl2_create", "cl2_destroy", "cl2_last_error", "cl2_abi_version", "cl2_build_bvh", "cl2_upload_scene", "cl2_set_seeds",
    "cl2_get_seeds", "cl2_make_light_rays", "cl2_make_camera_rays", "cl2_trace_light_rays",
    "cl2_trace_camera_rays", "cl2_join_paths", "cl2_finalize_samples", "cl2_gather_light_image",
    "cl2_process_images", "cl2_run_samples", "cl2_set_levels_per_launch", "cl2_set_traversal_mode", "cl2_set_pipelining", "cl2_read_accumulators", "cl2_reset_accumulators",
    "cl2_read_accumulators_packed", "cl2_write_accumulators_packed", "cl2_copy_accumulators_to_device",
    "cl2_copy_accumulators_from_device", "cl2_set_profiling", "cl2_set_counting", "cl2_set_debug_flags", "cl2_read_counters",
    "cl2_reset_counters", "cl2_selftest_exact_math", "cl2_export_rays", "cl2_export_paths", "cl2_export_aggregators",
    "cl2_export_sample_images", "cl2_probe_traverse", "cl2_probe_math", "cl2_probe_bounce",
]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in _SOURCES + [_HEADER])


def build(force=False, verbose=False):
    """Cross-compile the HIP library for gfx950 with hipcc (works without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = ["hipcc"] + HIPCC_FLAGS + [_SOURCES[0], "-o", LIB_PATH]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RendererError("hipcc failed:\n" + res.stdout)
    if verbose:
        print(" ".join(cmd))
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own `libamdhip64.so` (soname
    `libamdhip64.so.7`, found through torch's rpath), the library links `/opt/rocm`'s of the same
    soname.  Whichever is loaded first serves every later request for that SONAME, but torch asks by
    file name: if the system runtime came first, torch loads its own copy next to it, the second
    runtime finds no GPU, and device pointers could not be exchanged with `torch.distributed`
    anyway.  So when torch is installed, its runtime is loaded first (by path, without importing
    torch) and both sides share it.  CLIVE2_SYSTEM_HIP=1 keeps the system runtime (no torch interop)."""
    import sys
    if os.environ.get("CLIVE2_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass                                  # fall back to the runtime the library was linked against


def lib():
    """Load the library (never builds implicitly on a box without the sources' toolchain)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RendererError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.cl2_last_error.restype = C.c_char_p
        L.cl2_last_error.argtypes = [C.c_void_p]
        L.cl2_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.cl2_destroy.argtypes = [C.c_void_p]
        L.cl2_destroy.restype = None
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
