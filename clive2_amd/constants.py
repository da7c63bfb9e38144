"""Scene / camera / BVH constants of the reference (`src/constants.py`), same names.

Colours are in cv2 channel order (B, G, R) exactly as the reference defines them
(`src/constants.py:16-24`); the renderer never reorders channels.
"""
import time
import numpy as np

H_FOV = np.deg2rad(110.0)  # horizontal field of view, constants.py:5

UNIT_X, UNIT_Y, UNIT_Z = (np.eye(3, dtype=np.float64)[i] for i in range(3))
ZERO_VECTOR = np.zeros(3, dtype=np.float64)
INVALID = np.full(3, np.nan)
INF = np.full(3, np.inf)
NEG_INF = np.full(3, -np.inf)


def _bgr(b, g, r):
    return np.array([b, g, r], dtype=np.float64)


BLACK = _bgr(0.0, 0.0, 0.0)
WHITE = _bgr(0.7, 0.7, 0.7)
FULL_WHITE = _bgr(1.0, 1.0, 1.0)
GRAY = _bgr(0.5, 0.5, 0.5)
RED = _bgr(0.3, 0.3, 0.8)
GREEN = _bgr(0.541, 0.807, 0.0)
BLUE = _bgr(0.8, 0.3, 0.3)
CYAN = _bgr(0.8, 0.8, 0.3)

# BVH limits (constants.py:28-30)
MAX_MEMBERS = 8
MAX_DEPTH = 32
SPATIAL_SPLITS = 4

# Cornell box extents and light placement (constants.py:33-36)
DEFAULT_BOX_MIN_CORNER = np.array([-10, -2, -10])
DEFAULT_BOX_MAX_CORNER = np.array([10, 10, 10])
DEFAULT_LIGHT_HEIGHT = 0.95
DEFAULT_LIGHT_SCALE = 0.25

# path-tracing constants shared with the kernels (trace.metal:4-5, :407; renderer.py:8)
PI_F32 = np.float32(3.14159265359)
DELTA = np.float32(0.0001)
MAX_PATH_LENGTH = 8   # Path capacity
BOUNCE_LIMIT = 6      # hard-coded loop bound of generate_paths


def timed(func):
    """Print the wall time of each call (reference: constants.py:39-49).  Quiet unless
    CLIVE2_TIMED=1, because the reference's per-stage prints drown test output."""
    import functools
    import os

    @functools.wraps(func)
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return func(*a, **k)
        finally:
            if os.environ.get("CLIVE2_TIMED") == "1":
                print(f"Function {func.__name__} took {time.perf_counter() - t0:.4f} seconds")
    return wrapper
