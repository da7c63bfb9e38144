"""Native host BVH builder (cl2_build_bvh) against this package's numpy
restatement of the reference builder (itself byte-exact against the reference: test_scene_bvh.py).
No GPU needed."""
import numpy as np
import pytest

from clive2_amd import bvh, load
from clive2_amd.meshes import icosphere


def _random_soup(n, seed):
    """Independent random triangles: centroids are tie-free, so both builders must agree exactly."""
    rng = np.random.RandomState(seed)
    v = rng.uniform(-5, 5, size=(3 * n, 3))
    v[1::3] = v[0::3] + rng.normal(scale=0.3, size=(n, 3))
    v[2::3] = v[0::3] + rng.normal(scale=0.3, size=(n, 3))
    f = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    return v, f


@pytest.mark.parametrize("n,seed", [(9, 0), (100, 1), (1500, 2), (5000, 3)])
def test_native_equals_numpy_builder_on_tie_free_input(n, seed):
    v, f = _random_soup(n, seed)
    a, b = load.fast_load(v, f, material=3), load.fast_load(v, f, material=3)
    boxes_a, tris_a = bvh.np_flatten_bvh(bvh.construct_BVH(a, builder="numpy"))
    boxes_b, tris_b = bvh.np_flatten_bvh(bvh.construct_BVH(b, builder="native"))
    assert boxes_a.tobytes() == boxes_b.tobytes()
    assert tris_a.tobytes() == tris_b.tobytes()


def test_native_on_mesh_with_tied_centroids():
    """Meshes share vertices, so AABB centres tie; ties are ordered by triangle id instead of by
    numpy's quicksort.  The result must still be a valid tree over the same triangles."""
    v, f = icosphere(3, radius=2.0)
    a, b = load.fast_load(v, f, material=5), load.fast_load(v, f, material=5)
    boxes_a, tris_a = bvh.np_flatten_bvh(bvh.construct_BVH(a, builder="numpy"))
    boxes_b, tris_b = bvh.np_flatten_bvh(bvh.construct_BVH(b, builder="native"))
    for boxes, tris in ((boxes_a, tris_a), (boxes_b, tris_b)):
        seen = np.zeros(len(tris), int)
        for i, box in enumerate(boxes):
            if box["right"] == 0:
                assert i < box["left"] and box["left"] + 1 < len(boxes)
                for c in (box["left"], box["left"] + 1):
                    assert (boxes[c]["min"][:3] >= box["min"][:3]).all() and (boxes[c]["max"][:3] <= box["max"][:3]).all()
            else:
                seen[box["left"]:box["right"]] += 1
                t = tris[box["left"]:box["right"]]
                pts = np.stack([t["v0"], t["v1"], t["v2"]])[..., :3]
                assert (pts >= box["min"][:3] - 1e-6).all() and (pts <= box["max"][:3] + 1e-6).all()
                assert 1 <= box["right"] - box["left"] <= 8
        assert (seen == 1).all()
    key = lambda t: sorted(map(bytes, np.ascontiguousarray(t["v0"])))
    assert key(tris_a) == key(tris_b)                      # same triangle set
    # same quality: total node surface area within 2 %
    def total_area(bx):
        d = bx["max"][:, :3] - bx["min"][:, :3]
        return float(np.sum(2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])))
    assert abs(total_area(boxes_a) - total_area(boxes_b)) / total_area(boxes_a) < 0.02


def test_auto_builder_threshold_and_scene_validity():
    import clive2_amd as c2
    from clive2_amd.meshes import noisy_blob
    v, f = noisy_blob(subdiv=5)                              # 20,480 triangles > NATIVE_THRESHOLD
    s = c2.create_scene(32, 24, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=[dict(mesh=(v, f), material=5)])
    assert len(s.triangles) == 20480 + 16 and s.validate()
    assert s.triangles["is_light"].sum() == 2 and s.triangles["is_camera"].sum() == 2
    assert list(s.triangles["material"][s.light_triangle_indices]) == [6, 6]


def test_build_bvh_rejects_bad_arguments():
    import ctypes as C
    from clive2_amd import _native
    L = _native.lib()
    L.cl2_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                C.POINTER(C.c_int64), C.c_void_p]
    n_boxes = C.c_int64(0)
    assert L.cl2_build_bvh(None, None, 10, 8, 32, None, 0, C.byref(n_boxes), None) != 0
    L.cl2_last_error.restype = C.c_char_p
    assert b"cl2_build_bvh" in L.cl2_last_error(None)
