"""Nearest-first child order of the 4-wide walk (cl2_set_traversal_order(1); csrc/bvh_wide.hpp ORDER) against the exact walk.

    python tools/exp_order_ab.py [scene=glass,blob,interior] [samples=4] [W=1920] [H=1080]

Per scene, on ONE box:
  (i)   hit parity over all rays of `samples` samples of the pipeline: the subpath rays (every stored vertex of both Path[] buffers:
        origin + sampled direction) and the connection rays (light vertex s-1 -> camera vertex t-1, every (s, t >= 2) pair of a pixel),
        rebuilt on the host from the exact render's exported Path[] and sent through cl2_probe_traverse twice -- traversal mode 5 (the
        4-wide walk) with order 0 and with order 1.  A ray DIFFERS when (triangle, t bits) differ; a difference is a TIE when both
        walks report the same t bits (two triangles hit at exactly the same distance: the first one visited wins) and a NON-TIE
        otherwise (a hit in front of its own leaf box's entry distance, found or pruned depending on what was found before it);
  (ii)  what a ray costs in either order: wide-node visits, triangle records, own bytes (device tallies, cl2_set_counting(2)), ms per
        sample with 8 sample streams (pipelined) and the serial stage breakdown;
  (iii) the picture: per-pixel L2 distance between 64-spp renders with order 1 and order 0 (same seeds; the order-0 render is the one
        the parity suite compares byte for byte with the oracle), and the number of pixels above 1e-3.
Functions are imported by tests/test_gpu_fullsize.py (the >= 1e8-ray assertion of the GPU suite)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pipeline_ray_chunks(scene, samples=1, seed=0):
    """Yields (kind, origins (n,3) float32, directions (n,3) float32) for the rays of `samples` exact samples of the pipeline."""
    from clive2_amd.renderer import Renderer, make_seeds, LIGHT, CAMERA
    W, H = scene.pixel_width, scene.pixel_height
    r = Renderer(scene, seeds=make_seeds(W * H, seed=seed) if seed else make_seeds(W * H))
    try:
        for _ in range(samples):
            r.run_sample()
            verts = {}
            for which in (LIGHT, CAMERA):
                p = r.export_paths(which)
                length = p["length"].astype(np.int32)
                o = np.ascontiguousarray(p["rays"]["origin"][:, :6, :3])           # (B, 6, 3)
                d = np.ascontiguousarray(p["rays"]["direction"][:, :6, :3])
                del p
                verts[which] = (length, o)
                for k in range(6):
                    m = (length > k) & np.isfinite(d[:, k]).all(axis=1) & (np.abs(d[:, k]).sum(axis=1) > 0)
                    if m.any():
                        yield ("subpath", o[m, k], d[m, k])
                del d
            (len_l, o_l), (len_c, o_c) = verts[LIGHT], verts[CAMERA]
            for s in range(1, 7):
                for t in range(2, 7):
                    m = (len_l >= s) & (len_c >= t)
                    if not m.any():
                        continue
                    a, b = o_l[m, s - 1], o_c[m, t - 1]
                    v = (b - a).astype(np.float32)
                    n = np.sqrt((v * v).sum(axis=1, dtype=np.float32)).astype(np.float32)
                    ok = n > 0
                    yield ("connection", a[ok], (v[ok] / n[ok, None]).astype(np.float32))
    finally:
        r.close()


def leaf_entry_distance(scene, tri, o, d):
    """tmin of the slab test (trace.metal:150-156, the kernel's own float32 operations) of the LEAF box that holds triangle `tri`."""
    b = scene.boxes
    leaf = np.nonzero((b["right"] != 0) & (b["left"] <= tri) & (tri < b["right"]))[0]
    assert len(leaf) == 1, (tri, leaf)
    lo, hi = b["min"][leaf[0], :3].astype(np.float32), b["max"][leaf[0], :3].astype(np.float32)
    o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
    with np.errstate(divide="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
    t0, t1 = ((lo - o) * inv).astype(np.float32), ((hi - o) * inv).astype(np.float32)
    return float(max(np.minimum(t0, t1).max(), np.float32(0.0)))


def compare_orders(scene, chunks, log=None):
    """Sends every chunk through the 4-wide walk in both orders; returns the tallies of (i).  Every difference is explained:
    `in_front_of_own_leaf` counts the differing rays on which one of the two reported hits lies in front of its own leaf box's entry
    distance -- the one case in which the reference's result depends on its visit order (bvh_wide.hpp, ORDER); it must equal `differ`."""
    from clive2_amd import struct_types as st
    from clive2_amd.renderer import Renderer
    r0, r1 = Renderer(scene), Renderer(scene)
    out = {"rays": 0, "differ": 0, "ties": 0, "non_ties": 0, "in_front_of_own_leaf": 0, "missed_by_order1": 0, "missed_by_order0": 0, "by_kind": {},
           "examples": []}
    try:
        for r, order in ((r0, 0), (r1, 1)):
            r.set_traversal_mode(5)
            r.set_traversal_order(order)
        assert r0.traversal_order() == 0 and r1.traversal_order() == 1
        buf = None
        for kind, o, d in chunks:
            n = len(o)
            if buf is None or len(buf) < n:
                buf = np.zeros(n, dtype=st.Ray)
            rays = buf[:n]
            rays["origin"][:, :3] = o
            rays["direction"][:, :3] = d
            i0, t0, _, _ = r0.probe_traverse(rays)
            i1, t1, _, _ = r1.probe_traverse(rays)
            tb0, tb1 = t0.view(np.uint32), t1.view(np.uint32)
            diff = (i0 != i1) | (tb0 != tb1)
            nd = int(diff.sum())
            k = out["by_kind"].setdefault(kind, {"rays": 0, "differ": 0})
            k["rays"] += n; k["differ"] += nd
            out["rays"] += n; out["differ"] += nd
            if nd:
                idx = np.nonzero(diff)[0]
                tie = tb0[idx] == tb1[idx]
                out["ties"] += int(tie.sum()); out["non_ties"] += int((~tie).sum())
                out["missed_by_order1"] += int(((i1[idx] < 0) & (i0[idx] >= 0)).sum())
                out["missed_by_order0"] += int(((i0[idx] < 0) & (i1[idx] >= 0)).sum())
                for j in idx:
                    # the hit of either walk that lies in front of its own leaf box's entry distance (the one case in which a result
                    # depends on the visit order); a tie that survives the tie rule is such a hit too: the leaf of the triangle the
                    # reference met first is not entered once the other triangle is held at the same t
                    entry = [leaf_entry_distance(scene, int(i), o[j], d[j]) if i >= 0 else None for i in (i0[j], i1[j])]
                    front = [e is not None and float(t) < e for e, t in zip(entry, (t0[j], t1[j]))]
                    out["in_front_of_own_leaf"] += int(any(front))
                    if len(out["examples"]) < 32:
                        out["examples"].append({"kind": kind, "tie": bool(tb0[j] == tb1[j]), "tri": [int(i0[j]), int(i1[j])], "t": [float(t0[j]), float(t1[j])],
                                                "t_bits": ["%08x" % tb0[j], "%08x" % tb1[j]], "leaf_entry": entry, "in_front": front})
            if log:
                log(f"  {kind:10s} {n:9d} rays  differ {nd}")
    finally:
        r0.close(); r1.close()
    out["identical_fraction"] = 1.0 - out["differ"] / max(out["rays"], 1)
    return out


def walk_cost(scene, order, K=8, passes=3):
    """(ii): device tallies of the walk that runs + ms per sample (pipelined, K sample streams) + the serial stage breakdown."""
    from clive2_amd.renderer import Renderer, stream_seeds
    W, H = scene.pixel_width, scene.pixel_height
    r = Renderer(scene, seeds=stream_seeds(W * H, K), streams=K)
    try:
        r.set_traversal_order(order)
        r.tune()
        r.set_counting(2); r.reset_counters(); r.run_samples(1)
        t = r.walk_tallies()
        r.set_counting(False)
        r.run_samples(1); r.synchronize()
        t0 = time.perf_counter(); r.run_samples(passes); r.synchronize(); dt = time.perf_counter() - t0
        r.reset_counters(); r.set_profiling(2); r.set_pipelining(0); r.run_samples(1)
        c = r.counters()
        stages = {k[3:]: round(c[k] / K, 3) for k in c if k.startswith("ms_") and c[k] > 0}

        def per_ray(x):
            n = max(x["rays"], 1)
            return {"wide_visits": round(x["wide_visits"] / n, 3), "tri_records": round(x["tri_records"] / n, 3),
                    "own_bytes": round((112.0 * x["wide_visits"] + 36.0 * x["tri_records"] + 32.0 * x["binary_records"] + 16.0 * x["stack_spills"]) / n + 48.0, 1)}
        return {"order": order, "ms_per_sample": round(dt / (passes * K) * 1e3, 3), "serial_stage_ms_per_sample": stages,
                "connection": per_ray(t["connection"]), "subpath": per_ray(t["subpath"]), "rays_per_sample": c["rays"] // K,
                "paths_share": r.organisation()["paths_share"]}
    finally:
        r.close()


def picture_distance(scene, spp=64, K=8):
    """(iii): 64-spp renders in both orders on the same seeds; per-pixel L2 of the mean radiance."""
    from clive2_amd.renderer import Renderer, stream_seeds
    W, H = scene.pixel_width, scene.pixel_height
    imgs = []
    for order in (0, 1):
        r = Renderer(scene, seeds=stream_seeds(W * H, K), streams=K)
        r.set_traversal_order(order)
        r.set_reproducible(True)             # no atomics noise: what differs is the order alone
        r.run_samples(spp // K)
        imgs.append(np.asarray(r.radiance, dtype=np.float64).copy())      # summed_image / summed_sample_weights (renderer.py:295-297)
        r.close()
    d = np.sqrt(((imgs[0][..., :3] - imgs[1][..., :3]) ** 2).sum(axis=-1))
    return {"spp": spp, "pixels": int(d.size), "l2_max": float(d.max()), "l2_mean": float(d.mean()), "pixels_above_1e-3": int((d > 1e-3).sum()),
            "pixels_different_at_all": int((d > 0).sum())}


def main():
    import json
    import bench
    names = (sys.argv[1] if len(sys.argv) > 1 else "glass,blob,interior").split(",")
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
    for name in names:
        scene, desc = bench.build_scene(name, W, H)
        print(f"== {desc} {W}x{H}", flush=True)
        res = compare_orders(scene, pipeline_ray_chunks(scene, samples), log=lambda s: print(s, flush=True))
        print("hit parity:", json.dumps({k: v for k, v in res.items()}), flush=True)
        for order in (0, 1, 0, 1):
            print("cost:", json.dumps(walk_cost(scene, order)), flush=True)
        print("picture:", json.dumps(picture_distance(scene)), flush=True)
        bench._SCENES.clear()


if __name__ == "__main__":
    main()
