"""Scene geometry: Cornell box, camera quad, material table, mesh ingestion.

Same public names and array conventions as the reference's `src/load.py`
(`triangles_for_box` :203-258, `camera_geometry` :261-271, `get_materials` :179-200,
`fast_load` :98-134, `smooth_vertex_normals` :137-176, `surface_area` :274-277), written
against plain arrays.  `objloader` / `plyfile` (third-party, absent) are replaced by the
small readers in `meshio.py`.
"""
import numpy as np

from . import constants as C
from . import struct_types
from .bvh import FastTreeBox


def unit(v):
    return v / np.linalg.norm(v)


class Triangle:
    """One hand-placed triangle (Cornell walls, light, camera quad)."""

    __slots__ = ("v0", "v1", "v2", "n0", "n1", "n2", "n", "min", "max", "material",
                 "emitter", "camera", "surface_area")

    def __init__(self, v0, v1, v2, material=0, emitter=False, camera=False, normal=None):
        self.v0, self.v1, self.v2 = v0, v1, v2
        self.material, self.emitter, self.camera = material, emitter, camera
        self.min = np.minimum(v0, np.minimum(v1, v2))
        self.max = np.maximum(v0, np.maximum(v1, v2))
        self.n = unit(np.cross(v1 - v0, v2 - v0)) if normal is None else normal
        self.n0 = self.n1 = self.n2 = np.zeros(3)
        # The reference's constructor ends by overwriting the area with its (None) argument
        # (load.py:71-73); nothing downstream reads it, so keep a real value.
        self.surface_area = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0))


# Cornell corners are addressed by (x,y,z) bits: 0 = min side, 1 = max side.
_LBB, _RBB, _LTB, _LBF = (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)
_RTF, _LTF, _RBF, _RTB = (1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)

# (v0, v1, v2, material) in the reference's emission order (load.py:224-243); the vertex
# order fixes the barycentric parametrisation, so it is part of the contract.
_WALLS = [
    (_LBB, _RBB, _RTB, 4), (_LBB, _RTB, _LTB, 4),      # back   (z = min)
    (_LBB, _LTF, _LBF, 1), (_LBB, _LTB, _LTF, 1),      # left   (x = min)
    (_RBB, _RBF, _RTF, 2), (_RBB, _RTF, _RTB, 2),      # right  (x = max)
    (_LBF, _RTF, _RBF, 3), (_LBF, _LTF, _RTF, 3),      # front  (z = max, behind the camera)
    (_LBB, _RBF, _RBB, 4), (_LBB, _LBF, _RBF, 4),      # floor
    (_LTB, _RTB, _RTF, 4), (_LTB, _RTF, _LTF, 4),      # ceiling
]
_LIGHT = [(_LTB, _RTB, _RTF), (_LTB, _RTF, _LTF)]      # ceiling quad shrunk toward the axis


def triangles_for_box(box_min=C.DEFAULT_BOX_MIN_CORNER, box_max=C.DEFAULT_BOX_MAX_CORNER,
                      light_height=C.DEFAULT_LIGHT_HEIGHT, light_scale=C.DEFAULT_LIGHT_SCALE):
    """12 wall triangles + 2 emitter triangles (material 6).  The light is the ceiling quad
    scaled component-wise by (light_scale, light_height, light_scale) about the origin,
    which assumes a box centred in x/z (load.py:216-222, 244-257)."""
    span = np.asarray(box_max) - np.asarray(box_min)

    def corner(bits):
        return np.asarray(box_min) + span * np.array(bits, dtype=np.float64)

    shrink = np.array([light_scale, light_height, light_scale], dtype=np.float32)
    tris = [Triangle(corner(a), corner(b), corner(c), material=m) for a, b, c, m in _WALLS]
    tris += [Triangle(corner(a) * shrink, corner(b) * shrink, corner(c) * shrink,
                      material=6, emitter=True) for a, b, c in _LIGHT]
    return tris


def camera_geometry(camera):
    """The film rectangle as two `is_camera` triangles (material 7), so light subpaths can
    be projected onto it by ordinary ray casting (trace.metal:569-617)."""
    o = camera.origin
    right = camera.dx * camera.phys_width
    up = camera.dy * camera.phys_height
    return [Triangle(o, o + right, o + right + up, material=7, camera=True),
            Triangle(o, o + right + up, o + up, material=7, camera=True)]


def get_materials():
    """The 8-entry material table (load.py:179-200): all alpha 0, ior 1.5; types 0/5 are
    smooth dielectrics (type 1), the rest Lambertian (type 0); entry 6 emits (1,1,1,1)."""
    m = np.zeros(8, dtype=struct_types.Material)
    for i, col in enumerate((C.RED, C.GREEN, C.BLUE, C.WHITE, C.WHITE, C.BLUE,
                             C.FULL_WHITE, C.FULL_WHITE)):
        m["color"][i, :3] = col
    m["emission"][6] = 1.0
    m["ior"] = 1.5
    m["alpha"] = 0.0
    m["type"][[0, 5]] = 1
    return m


def smooth_vertex_normals(vertices, faces, face_n):
    """Angle-weighted vertex normals.  `face_n` is used as given: `fast_load` passes the
    UN-normalised cross products, so the effective weight is angle x 2*area
    (load.py:105-109, 137-176)."""
    corners = vertices[faces]                                   # (M,3,3)
    to_next = np.roll(corners, -1, axis=1) - corners
    to_prev = np.roll(corners, 1, axis=1) - corners
    angle = np.arctan2(np.linalg.norm(np.cross(to_next, to_prev), axis=2),
                       np.einsum("fck,fck->fc", to_next, to_prev))   # (M,3); einsum's
    # accumulation order is the defining one (load.py:162): a plain .sum() differs in the last bit
    contrib = (face_n[:, None, :] * angle[..., None]).reshape(-1, 3)
    # np.add.at is the defining accumulation order (load.py:169-170); a bincount-based sum
    # differs in the last bit, which would break the byte-exact fixture parity.
    acc = np.zeros_like(vertices, dtype=vertices.dtype)
    np.add.at(acc, faces.ravel(), contrib)
    length = np.linalg.norm(acc, axis=1, keepdims=True)
    np.divide(acc, length, out=acc, where=length > 0)
    return acc


def fast_load(vertices, faces, emitter=False, material=None):
    """Indexed mesh -> triangle soup with smooth shading normals (load.py:98-134)."""
    tri = vertices[faces]
    raw_n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    vertex_n = smooth_vertex_normals(vertices, faces, raw_n)
    twice_area = np.linalg.norm(raw_n, axis=1)
    n = len(tri)
    return FastTreeBox(
        faces=faces, triangles=tri, mins=tri.min(axis=1), maxes=tri.max(axis=1),
        face_normals=raw_n / twice_area[:, None], smoothed_normals=vertex_n[faces],
        surface_areas=twice_area / 2,
        material=np.full(n, 0 if material is None else material, dtype=np.int32),
        emitter=np.full(n, bool(emitter), dtype=np.bool_),
        camera=np.zeros(n, dtype=np.int32),
    )


def fast_load_obj(obj_path, offset=None, material=None, emitter=False, scale=1.0):
    from .meshio import read_obj
    v, f = read_obj(obj_path)
    v = v * scale + (np.zeros(3) if offset is None else offset)
    return fast_load(v, f, material=material, emitter=emitter)


def fast_load_ply(ply_path, offset=None, material=None, scale=1.0, emitter=False):
    from .meshio import read_ply
    v, f = read_ply(ply_path)          # float32 xyz, as the reference views them (load.py:91-93)
    v = v * scale + (np.zeros(3) if offset is None else offset)
    return fast_load(v, f, material=material, emitter=emitter)


def surface_area(t):
    """Area of one flattened Triangle record (load.py:274-277)."""
    return np.linalg.norm(np.cross((t["v1"] - t["v0"])[:3], (t["v2"] - t["v0"])[:3])) / 2
