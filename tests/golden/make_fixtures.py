"""Generate golden fixtures from the reference's HOST-side Python (scene/BVH/camera plumbing and
the numpy host glue of renderer.py: light_bins, process_images, the image properties).

Runs ONLY in the authoring container (needs /root/reference, which never travels to the
GPU box).  The device code of the reference (trace.metal) cannot run anywhere in this
pipeline, so these fixtures pin SURVEY.md §8 row a21 only (scene -> Box[]/Triangle[]/
Material[]/Camera[] arrays).  The reference modules are imported unmodified; the four
third-party modules they import that are absent here are replaced by inert in-memory
stand-ins (numba.njit = identity, metalcompute.Device.buffer(x) = x; objloader/plyfile
are never called).  Output: small .npz files with raw struct bytes.

    python tests/golden/make_fixtures.py
"""
import io
import os
import sys
import types
import contextlib

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stand_ins():
    numba = types.ModuleType("numba")
    numba.njit = lambda f=None, **kw: f if f is not None else (lambda g: g)
    sys.modules["numba"] = numba
    sys.modules["objloader"] = types.ModuleType("objloader")
    ply = types.ModuleType("plyfile")
    ply.PlyData = object
    sys.modules["plyfile"] = ply
    mc = types.ModuleType("metalcompute")

    class _Kernel:                       # dev.kernel(text).function(name): the Metal JIT, inert here
        def function(self, name):
            return lambda *a, **k: None

    class Device:
        def buffer(self, x):             # nbytes -> zeroed buffer-protocol object; ndarray -> itself
            if isinstance(x, (int, np.integer)):
                return np.zeros(int(x), dtype=np.uint8)
            return x

        def kernel(self, text):
            return _Kernel()

    mc.Device = Device
    mc.release = lambda *_a, **_k: None
    mc.error = RuntimeError
    sys.modules["metalcompute"] = mc


def _raw(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(a.tobytes(), dtype=np.uint8).copy()


def scene_fixture(ref_scene, w, h):
    with contextlib.redirect_stdout(io.StringIO()):
        s = ref_scene.create_scene_from_preset("empty", w, h)
    out = dict(
        boxes=_raw(s.boxes), triangles=_raw(s.triangles), materials=_raw(s.materials),
        camera=_raw(s.camera), light_triangles=_raw(s.light_triangles),
        light_surface_areas=np.asarray(s.light_surface_areas, dtype=np.float32),
        light_triangle_indices=np.asarray(s.light_triangle_indices, dtype=np.int32),
        camera_triangle_indices=np.asarray(s.camera_triangle_indices, dtype=np.int32),
        light_counts=np.asarray(s.light_counts, dtype=np.int32).reshape(1),
        n_boxes=np.int32(len(s.boxes)), n_triangles=np.int32(len(s.triangles)),
    )
    s.__class__.__del__ = lambda self: None
    return out


def icosphere(subdiv):
    """Deterministic icosphere (unit radius): the same generator the package ships."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int32)


def mesh_fixture(ref_scene, ref_load, ref_bvh, ref_camera, subdiv, w, h):
    """Cornell box + one smooth-shaded icosphere pushed through the reference's
    fast_load -> construct_BVH -> np_flatten_bvh chain (scene.py:41-71)."""
    verts, faces = icosphere(subdiv)
    verts = verts * 2.0 + np.array([0.0, 1.0, 0.0])
    with contextlib.redirect_stdout(io.StringIO()):
        cam = ref_camera.Camera(center=np.array([0, 1.5, 6]), direction=np.array([0, 0, -1]),
                                pixel_width=w, pixel_height=h, phys_width=w / h, phys_height=1)
        tris = list(ref_load.camera_geometry(cam)) + list(ref_load.triangles_for_box())
        box = ref_bvh.FastTreeBox.from_triangle_objects(tris)
        box = box + ref_load.fast_load(verts, faces, material=5)
        bvh = ref_bvh.construct_BVH(box)
        np_boxes, np_tris = ref_bvh.np_flatten_bvh(bvh)
    return dict(vertices=verts, faces=faces, boxes=_raw(np_boxes), triangles=_raw(np_tris),
                n_boxes=np.int32(len(np_boxes)), n_triangles=np.int32(len(np_tris)))


def spatial_fixture(ref_load, ref_bvh, ref_camera, w, h):
    """Outputs of the reference's `spatial_split` (bvh.py:194-285; dead code there: its call in construct_BVH is
    commented out, bvh.py:298-299) on a few soups: the Cornell box + an icosphere, and the two children of its first
    object split.  Stored: the soup (vertices/faces to rebuild it), the returned cost and the children's triangles."""
    verts, faces = icosphere(1)
    verts = verts * 2.0 + np.array([0.0, 1.0, 0.0])
    out = dict(vertices=verts, faces=faces)
    with contextlib.redirect_stdout(io.StringIO()):
        cam = ref_camera.Camera(center=np.array([0, 1.5, 6]), direction=np.array([0, 0, -1]),
                                pixel_width=w, pixel_height=h, phys_width=w / h, phys_height=1)
        tris = list(ref_load.camera_geometry(cam)) + list(ref_load.triangles_for_box())
        root = ref_bvh.FastTreeBox.from_triangle_objects(tris) + ref_load.fast_load(verts, faces, material=5)
        _, obj_l, obj_r = ref_bvh.object_split(root)
        room = ref_bvh.FastTreeBox.from_triangle_objects(tris)
        for name, box in (("root", root), ("left", obj_l), ("right", obj_r), ("room", room)):
            cost, l, r = ref_bvh.spatial_split(box)
            out[name + "_in"] = np.asarray(box.triangles)
            out[name + "_cost"] = np.float64(cost)
            out[name + "_l"] = np.asarray(l.triangles) if l is not None else np.zeros((0, 3, 3))
            out[name + "_r"] = np.asarray(r.triangles) if r is not None else np.zeros((0, 3, 3))
    return out


def renderer_glue_fixture(ref_scene, ref_renderer, w, h, seed):
    """The numpy host glue of the reference's Renderer (renderer.py:11-13, :63, :97-111, :253-316) run on
    seeded stand-ins for the device buffers: light-image indices with negative (unused) and >= W*H
    (off-frame) entries, per-sample images with NaN / +-inf entries.  Inputs and outputs are
    stored; the device kernels are never called (the fake metalcompute's functions do nothing)."""
    rng = np.random.RandomState(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        s = ref_scene.create_scene_from_preset("empty", w, h)
        r = ref_renderer.Renderer(s, kernel_path=os.path.join(REF, "trace.metal"))
    s.__class__.__del__ = lambda self: None
    r.__class__.__del__ = lambda self: None
    B = w * h
    n = ref_renderer.next_power_of_two(B * ref_renderer.MAX_PATH_LENGTH)
    assert len(r.out_light_indices) == 4 * n
    idx = rng.randint(-1, B + 3, size=n).astype(np.int32)
    idx[rng.rand(n) < 0.5] = -1
    r.out_light_indices = idx.view(np.uint8)
    bins, offset = r.light_bins()
    out = dict(width=np.int32(w), height=np.int32(h), n=np.int64(n), light_indices=idx,
               summed_bins=np.asarray(bins).copy(), offset=np.asarray(offset).copy(),
               npot_in=np.array([0, 1, 2, 3, 5, 8, 1000, 65536 * 8, 1920 * 1080 * 8], np.int64))
    out["npot_out"] = np.array([ref_renderer.next_power_of_two(int(v)) for v in out["npot_in"]], np.int64)

    def spoil(a):                        # NaN / inf entries, as broken paths produce them
        m = rng.rand(*a.shape)
        a[m < 0.02] = np.nan
        a[(m >= 0.02) & (m < 0.03)] = np.inf
        a[(m >= 0.03) & (m < 0.04)] = -np.inf
        return a

    samples = []
    for _ in range(3):
        fin = spoil((rng.rand(B, 4) * 2.0).astype(np.float32))
        light = spoil((rng.rand(B, 4) * 0.5).astype(np.float32))
        uni = spoil((rng.rand(B, 4) * 3.0).astype(np.float32))
        cnt = np.ones(B, np.int32)
        wts = (rng.rand(B) * 9.0).astype(np.float32)
        wts[rng.rand(B) < 0.05] = 0.0    # zero weight mass -> 0/0 and x/0 in the final ratio
        r.finalized_samples, r.out_light_image, r.out_camera_image = fin.view(np.uint8), light.view(np.uint8), uni.view(np.uint8)
        r.sample_counts, r.sample_weights = cnt.view(np.uint8), wts.view(np.uint8)
        r.process_images()
        samples.append((fin, light, uni, cnt, wts))
    for k, name in enumerate(("finalized", "light", "unidirectional", "counts", "weights")):
        out["in_" + name] = np.stack([smp[k] for smp in samples])
    out.update(summed_image=r.summed_image.copy(), summed_sample_counts=r.summed_sample_counts.copy(),
               summed_sample_weights=r.summed_sample_weights.copy(),
               unidirectional_image_buffer=r.unidirectional_image_buffer.copy())
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        out.update(image=r.image, unweighted_image=r.unweighted_image, unidirectional_image=r.unidirectional_image)
    return out


def main():
    _install_stand_ins()
    sys.path.insert(0, REF)
    import scene as ref_scene
    import load as ref_load
    import bvh as ref_bvh
    import camera as ref_camera
    import struct_types as ref_st
    import renderer as ref_renderer

    sizes = {n: np.int32(getattr(ref_st, n).itemsize)
             for n in ("Ray", "Path", "Box", "Triangle", "Material", "Camera")}
    offsets = {}
    for n in ("Ray", "Path", "Box", "Triangle", "Material", "Camera"):
        dt = getattr(ref_st, n)
        for fname in dt.names:
            offsets[f"{n}.{fname}"] = np.int32(dt.fields[fname][1])
    np.savez_compressed(os.path.join(OUT, "struct_layout.npz"), **sizes, **offsets)

    for (w, h) in ((256, 256), (1920, 1080), (64, 48)):
        np.savez_compressed(os.path.join(OUT, f"cornell_{w}x{h}.npz"), **scene_fixture(ref_scene, w, h))

    for subdiv in (1, 2):
        np.savez_compressed(os.path.join(OUT, f"cornell_icosphere{subdiv}_64x48.npz"),
                            **mesh_fixture(ref_scene, ref_load, ref_bvh, ref_camera, subdiv, 64, 48))

    np.savez_compressed(os.path.join(OUT, "spatial_split.npz"), **spatial_fixture(ref_load, ref_bvh, ref_camera, 64, 48))

    # tone_map (camera.py:73-82) on a small seeded image: output-stage fixture
    rng = np.random.RandomState(7)
    img = rng.rand(12, 16, 3).astype(np.float32) * 3.0
    with contextlib.redirect_stdout(io.StringIO()):
        tm = ref_camera.tone_map(img, exposure=4.0)
    np.savez_compressed(os.path.join(OUT, "tone_map.npz"), image=img, out=tm)
    np.savez_compressed(os.path.join(OUT, "renderer_glue.npz"), **renderer_glue_fixture(ref_scene, ref_renderer, 24, 16, 11))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
