"""Thin-film + pinhole camera and the display tone map (reference behaviour: `src/camera.py`).

Film = the `phys_width x phys_height` rectangle centred on `center` and spanned by the unit
vectors `dx`, `dy`; every camera ray passes through `focal_point`, which lies `focal_dist` along
`direction` (`camera.py:34-39`).  `to_struct()` fills the 112-byte Camera record of the kernels.
"""
import numpy as np

from .constants import H_FOV, UNIT_X, UNIT_Y, UNIT_Z
from . import struct_types

_EPS = 0.0001
_RECORD_VECTORS = ("center", "focal_point", "direction", "dx", "dy")
_RECORD_SCALARS = ("pixel_width", "pixel_height", "phys_width", "phys_height", "h_fov", "v_fov")


def film_basis(direction):
    """(dx, dy) for a viewing direction, with the reference's special cases: a camera looking along
    +-z gets dx = +-x (`camera.py:42-48`), a level camera gets dy = +y (`:50-55`)."""
    d = np.asarray(direction, dtype=np.float64)
    if abs(d[0]) < _EPS:
        dx = UNIT_X if d[2] > 0 else -UNIT_X
    else:
        dx = np.cross(d * (UNIT_X + UNIT_Z), -UNIT_Y)
        dx = dx / np.linalg.norm(dx)
    if abs(d[1]) < _EPS:
        dy = UNIT_Y
    else:
        dy = np.cross(d, dx)
        dy = dy / np.linalg.norm(dy)
    return dx, dy


class Camera:
    def __init__(self, center=np.zeros(3), direction=np.array([1, 0, 0]), phys_width=1.0,
                 phys_height=1.0, pixel_width=1280, pixel_height=720):
        self.center, self.direction = np.asarray(center), np.asarray(direction)
        self.phys_width, self.phys_height = phys_width, phys_height
        self.pixel_width, self.pixel_height = pixel_width, pixel_height
        self.aspect_ratio = phys_width / phys_height
        self.h_fov = H_FOV
        self.v_fov = 2.0 * np.arctan(np.tan(0.5 * H_FOV) / self.aspect_ratio)
        self.dx, self.dy = film_basis(self.direction)
        self.focal_dist = phys_width / (2 * np.tan(0.5 * self.h_fov))
        self.focal_point = self.center + self.focal_dist * self.direction
        # per-pixel film steps, pixel area, and the film's lower corner (`camera.py:27-31`)
        self.dx_dp = self.dx * (phys_width / pixel_width)
        self.dy_dp = self.dy * (phys_height / pixel_height)
        self.pixel_phys_size = np.linalg.norm(self.dx_dp) * np.linalg.norm(self.dy_dp)
        self.origin = self.center - self.dx * phys_width / 2 - self.dy * phys_height / 2

    def to_struct(self):
        rec = np.zeros(1, dtype=struct_types.Camera)
        for name in _RECORD_VECTORS:
            rec[name][0, :3] = getattr(self, name)
        for name in _RECORD_SCALARS:
            rec[name][0] = getattr(self, name)
        return rec


_LUMA_BGR = np.array([0.0722, 0.7152, 0.2126])


def tone_map(image, exposure=2.0, white_point=1.0, verbose=False):
    """Log-average (Reinhard-style) operator -> uint8, BGR (`camera.py:73-82`): scale by
    exposure / exp(mean log(0.1 + luma)), then compress x / (x + white^2)."""
    if verbose:
        print(f"IN min: {np.min(image)}, mean: {np.mean(image)}, max: {np.max(image)}")
    luma = (image * _LUMA_BGR).sum(axis=2)
    log_avg = np.exp(np.log(0.1 + luma).sum() / (image.shape[0] * image.shape[1]))
    scaled = image * exposure / log_avg
    return (255 * scaled / (scaled + white_point ** 2)).astype(np.uint8)
