"""Thin-film + pinhole camera model and the display tone map (reference: `src/camera.py`).

The film is a `phys_width x phys_height` rectangle centred on `center`, spanned by
`dx`,`dy`; every camera ray passes through `focal_point = center + focal_dist*direction`
(`camera.py:34-39`).  `to_struct()` fills the 112-byte Camera record consumed by the
kernels.
"""
import numpy as np

from .constants import H_FOV, UNIT_X, UNIT_Y, UNIT_Z
from . import struct_types


class Camera:
    def __init__(self, center=np.zeros(3), direction=np.array([1, 0, 0]), phys_width=1.0,
                 phys_height=1.0, pixel_width=1280, pixel_height=720):
        self.center = np.asarray(center)
        self.direction = np.asarray(direction)
        self.phys_width, self.phys_height = phys_width, phys_height
        self.pixel_width, self.pixel_height = pixel_width, pixel_height
        self.aspect_ratio = phys_width / phys_height
        self.h_fov = H_FOV
        self.v_fov = 2.0 * np.arctan(np.tan(0.5 * H_FOV) / self.aspect_ratio)
        # per-pixel film steps and the film's lower corner (camera.py:27-31)
        self.dx_dp = self.dx * (phys_width / pixel_width)
        self.dy_dp = self.dy * (phys_height / pixel_height)
        self.pixel_phys_size = np.linalg.norm(self.dx_dp) * np.linalg.norm(self.dy_dp)
        self.origin = self.center - self.dx * phys_width / 2 - self.dy * phys_height / 2

    @property
    def focal_dist(self):
        return self.phys_width / (2 * np.tan(self.h_fov / 2))

    @property
    def focal_point(self):
        return self.center + self.focal_dist * self.direction

    @property
    def dx(self):
        d = self.direction
        if abs(d[0]) < 0.0001:                     # looking along +-z: camera.py:43-44
            return UNIT_X if d[2] > 0 else -UNIT_X
        v = np.cross(d * (UNIT_X + UNIT_Z), -UNIT_Y)
        return v / np.linalg.norm(v)

    @property
    def dy(self):
        d = self.direction
        if abs(d[1]) < 0.0001:
            return UNIT_Y
        v = np.cross(d, self.dx)
        return v / np.linalg.norm(v)

    def to_struct(self):
        rec = np.zeros(1, dtype=struct_types.Camera)
        for name in ("center", "focal_point", "direction", "dx", "dy"):
            rec[name][0, :3] = getattr(self, name)
        for name in ("pixel_width", "pixel_height", "phys_width", "phys_height", "h_fov", "v_fov"):
            rec[name][0] = getattr(self, name)
        return rec


_LUMA_BGR = np.array([0.0722, 0.7152, 0.2126])


def tone_map(image, exposure=2.0, white_point=1.0, verbose=False):
    """Log-average (Reinhard-style) operator -> uint8, BGR (reference: camera.py:73-82)."""
    if verbose:
        print(f"IN min: {np.min(image)}, mean: {np.mean(image)}, max: {np.max(image)}")
    luma = (image * _LUMA_BGR).sum(axis=2)
    log_avg = np.exp(np.log(0.1 + luma).sum() / (image.shape[0] * image.shape[1]))
    scaled = image * exposure / log_avg
    return (255 * scaled / (scaled + white_point ** 2)).astype(np.uint8)
