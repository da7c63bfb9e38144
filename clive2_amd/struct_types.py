"""Byte-exact host<->device record layouts (the ABI contract of the drop-in boundary).

Mirrors the six numpy structured dtypes of the reference (`src/struct_types.py:4-85`)
and the Metal struct declarations they shadow (`src/trace.metal:7-85`): every float3 is
padded to 16 B, records are AoS.  The layouts are written here as explicit
(name, format, offset) tables so a mismatch with the C headers in `include/` is a
one-line diff; `tests/test_struct_layout.py` pins them against the golden layout
captured from the reference (`tests/golden/struct_layout.npz`).

The library keeps these layouts only AT THE BOUNDARY (scene upload, debug export);
device-side state is SoA.
"""
import numpy as np

_F3 = (np.float32, (4,))  # float3 padded to 16 bytes
_I = np.int32
_F = np.float32


def _record(size, fields):
    names, formats, offsets = zip(*fields)
    return np.dtype({"names": list(names), "formats": list(formats),
                     "offsets": list(offsets), "itemsize": size})


# struct Ray, 128 B (trace.metal:7-23)
Ray = _record(128, [
    ("origin", _F3, 0), ("direction", _F3, 16), ("inv_direction", _F3, 32),
    ("color", _F3, 48), ("normal", _F3, 64),
    ("material", _I, 80), ("triangle", _I, 84),
    ("c_importance", _F, 88), ("l_importance", _F, 92), ("tot_importance", _F, 96),
    ("hit_light", _I, 100), ("from_camera", _I, 104), ("hit_camera", _I, 108),
    ("pixel_idx", _I, 112), ("pad", (_I, (3,)), 116),
])

# struct Path, 1040 B (trace.metal:31-36): 8 vertex slots, 6 ever used (trace.metal:407)
Path = _record(1040, [
    ("rays", (Ray, (8,)), 0), ("length", _I, 1024), ("from_camera", _I, 1028),
    ("pad", (_I, (2,)), 1032),
])

# struct Box, 48 B (trace.metal:38-44): right==0 -> inner (children left,left+1) else leaf [left,right)
Box = _record(48, [
    ("min", _F3, 0), ("max", _F3, 16), ("left", _I, 32), ("right", _I, 36),
    ("pad", (_I, (2,)), 40),
])

# struct Triangle, 128 B (trace.metal:47-59)
Triangle = _record(128, [
    ("v0", _F3, 0), ("v1", _F3, 16), ("v2", _F3, 32),
    ("n0", _F3, 48), ("n1", _F3, 64), ("n2", _F3, 80), ("normal", _F3, 96),
    ("material", _I, 112), ("is_light", _I, 116), ("is_camera", _I, 120), ("pad", _I, 124),
])

# struct Material, 48 B (trace.metal:62-69)
Material = _record(48, [
    ("color", _F3, 0), ("emission", _F3, 16), ("type", _I, 32), ("alpha", _F, 36),
    ("ior", _F, 40), ("pad", _I, 44),
])

# struct Camera, 112 B (trace.metal:72-85)
Camera = _record(112, [
    ("center", _F3, 0), ("focal_point", _F3, 16), ("direction", _F3, 32),
    ("dx", _F3, 48), ("dy", _F3, 64),
    ("pixel_width", _I, 80), ("pixel_height", _I, 84),
    ("phys_width", _F, 88), ("phys_height", _F, 92), ("h_fov", _F, 96), ("v_fov", _F, 100),
    ("pad", (_I, (2,)), 104),
])

# struct WeightAggregator (device-only in the reference, trace.metal:25-29): float[3][3] @0,
# float3 total_contribution @48 (16-B aligned), float contrib_weight_sum @64; the reference host
# allocates 128 B per element (renderer.py:71), which is the export stride used here.
WeightAggregator = _record(128, [
    ("weights", (_F, (3, 3)), 0), ("total_contribution", _F3, 48),
    ("contrib_weight_sum", _F, 64),
])

ALL = {"Ray": Ray, "Path": Path, "Box": Box, "Triangle": Triangle,
       "Material": Material, "Camera": Camera}
