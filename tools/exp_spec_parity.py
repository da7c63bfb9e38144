"""Parity of the 4-wide walk WITHOUT the speculative stack-top expansion (debug bit 13 switches it off; the default, flags 0, is the
speculative walk) against the oracle (round 5's log profiles/r05_spec_parity.log was taken with 0x6000 on the build of
profiles/r05_spec_pop_between_rounds.patch, whose second-level speculation sat behind bit 14: that bit no longer exists): small glass scene in modes 0 / 5, the 1080p
config-3 frame (one sample + two pipelined), the 20k-triangle blob at 128x72: Path[] and aggregators byte for byte."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import clive2_amd as c2
from clive2_amd.renderer import Renderer, make_seeds
from clive2_amd.load import get_materials
from clive2_amd.meshes import icosphere, noisy_blob
from oracle import oracle as orc
orc.build()
FLAGS = int(sys.argv[1], 0) if len(sys.argv) > 1 else (1 << 13)

def scene_glass(sub, w, h):
    mats = get_materials(); mats["alpha"][5] = 0.1
    v, f = icosphere(sub, radius=2.0, center=(0.0, 1.0, 0.0))
    return c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=[dict(mesh=(v, f), material=5)], materials=mats)

def scene_blob(w, h):
    mats = get_materials(); mats["alpha"][5] = 0.1
    v, f = noisy_blob(subdiv=5)
    return c2.create_scene(w, h, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=[dict(mesh=(v, f), material=5)], materials=mats)

def check(scene, mode, n_more, tag):
    B = scene.pixel_width * scene.pixel_height
    seeds = make_seeds(B)
    r, o = Renderer(scene, seeds=seeds), orc.OracleRenderer(scene, seeds=seeds)
    r.set_traversal_mode(mode); r.set_debug_flags(FLAGS)
    r.run_samples(1); o.run_sample()
    for k in range(n_more):
        r.run_samples(2); o.run_sample(); o.run_sample()
    ok = (r.export_paths(0).tobytes() == o.out_light_paths.tobytes() and r.export_paths(1).tobytes() == o.out_camera_paths.tobytes()
          and r.export_aggregators()["total_contribution"].tobytes() == o.weight_aggregators["total_contribution"].tobytes()
          and np.array_equal(r.get_random_buffer(), o.rand_buffer) and r.counters()["rays"] == o.rays_traced)
    print(tag, "mode", mode, "flags %#x" % FLAGS, "OK" if ok else "MISMATCH", flush=True)
    r.close()
    return ok

good = True
g = scene_glass(2, 64, 48)
for mode in (5, 0):
    good &= check(g, mode, 1, "glass sub2 64x48")
good &= check(scene_glass(4, 160, 90), 0, 1, "glass sub4 160x90")
good &= check(scene_blob(128, 72), 0, 1, "blob sub5 128x72")
good &= check(scene_glass(4, 1920, 1080), 0, 1, "glass sub4 1080p")
sys.exit(0 if good else 1)
