"""(Needs profiles/r06_two_lanes_per_ray.patch applied.)  Parity of the two-lanes-per-ray walk (debug bit 15, csrc/bvh_wide2.hpp) against the ORACLE before it is timed:
    [CL2_LIB=build/lib_x.so] python tools/exp_pairs_parity.py
config-3 geometry (modes 5 and 0, speculation on and off), the 20k-triangle blob, an open scene, a ragged frame, the 1M-triangle
interior (the streaming form: one pair round per pass), config 3 at 1080p (one serial + two pipelined samples, 4 sample streams),
and 200,000 probe rays with zero direction components (non-finite 1/d: the binary records inside the same kernel) against the
one-lane-per-ray walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clive2_amd._native as _n
if os.environ.get("CL2_LIB"):
    _n.LIB_PATH = os.path.abspath(os.environ["CL2_LIB"])
import bench
from clive2_amd import struct_types as st
from clive2_amd.renderer import Renderer, make_seeds, stream_seeds
from oracle import oracle as orc
orc.build()
PAIRS = 1 << 15
good = True


def check(scene, tag, mode, flags, K=1, more=1):
    global good
    B = scene.pixel_width * scene.pixel_height
    seeds = stream_seeds(B, K)
    r = Renderer(scene, seeds=seeds, streams=K)
    os_ = [orc.OracleRenderer(scene, seeds=seeds if K == 1 else seeds[j]) for j in range(K)]
    r.set_traversal_mode(mode); r.set_debug_flags(flags)
    r.run_samples(1)
    for o in os_: o.run_sample()
    for _ in range(more):
        r.run_samples(2)
        for o in os_: o.run_sample(); o.run_sample()
    ok = True
    for j, o in enumerate(os_):
        r.set_export_stream(j)
        ok &= r.export_paths(0).tobytes() == o.out_light_paths.tobytes() and r.export_paths(1).tobytes() == o.out_camera_paths.tobytes()
        agg = r.export_aggregators()
        ok &= all(agg[f].tobytes() == o.weight_aggregators[f].tobytes() for f in ("weights", "total_contribution", "contrib_weight_sum"))
    ok &= r.counters()["rays"] == sum(o.rays_traced for o in os_)
    print(f"{tag:28s} mode {mode} flags {flags:#x} K {K}: {'OK' if ok else 'MISMATCH'}", flush=True)
    good &= bool(ok)
    r.close()


for name, W, H in (("glass", 160, 90), ("blob", 128, 72), ("open", 91, 60), ("open", 257, 1)):
    scene, _ = bench._build_scene(name, W, H)
    for mode, flags in ((5, PAIRS), (0, PAIRS), (5, PAIRS | (1 << 13)), (5, PAIRS | (1 << 20))):
        check(scene, f"{name} {W}x{H}", mode, flags)
scene, _ = bench._build_scene("glass", 131, 77)
check(scene, "glass 131x77 (ragged)", 5, PAIRS, K=3)
scene, _ = bench._build_scene("interior", 96, 54)
for mode, flags in ((5, PAIRS), (0, PAIRS), (5, PAIRS | (1 << 13))):
    check(scene, "interior 96x54", mode, flags)
# rays with zero direction components through both walks
rng = np.random.RandomState(7)
n = 200000
rays = np.zeros(n, dtype=st.Ray)
rays["origin"][:, :3] = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
dd = rng.normal(size=(n, 3)).astype(np.float32)
dd[np.arange(n), rng.randint(0, 3, n)] = 0.0
dd[: n // 4, rng.randint(0, 3)] = 0.0
dd /= np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-20)
rays["direction"][:, :3] = dd
hits = []
for flags in (0, PAIRS):
    r = Renderer(scene)
    r.set_traversal_mode(5); r.set_debug_flags(flags)
    hits.append(r.probe_traverse(rays))
    r.close()
same = all(a.tobytes() == b.tobytes() for a, b in zip(hits[0], hits[1]))
print("probe rays with zero direction components, interior:", "OK" if same else "MISMATCH", int((hits[0][0] >= 0).sum()), "hits", flush=True)
good &= same
scene, _ = bench._build_scene("glass", 1920, 1080)
check(scene, "glass 1920x1080", 0, PAIRS, K=1)
check(scene, "glass 1920x1080", 0, PAIRS, K=4, more=1)
sys.exit(0 if good else 1)
