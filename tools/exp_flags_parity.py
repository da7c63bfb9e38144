"""Parity of a debug-flag setting against the oracle on a handful of scenes (Cornell box, rough glass, all material types, an open
scene with short subpaths, a ragged frame), serial call + pipelined calls:   python tools/exp_flags_parity.py <flags>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import clive2_amd as c2
from clive2_amd.renderer import Renderer, make_seeds
from oracle import oracle as orc
orc.build()
FLAGS = int(sys.argv[1], 0)
good = True
for name, W, H in (("cornell", 64, 48), ("cornell", 1920, 1080), ("glass", 160, 90), ("open", 91, 60), ("open", 257, 1), ("blob", 128, 72)):
    scene, _ = bench._build_scene(name, W, H)
    seeds = make_seeds(W * H)
    r, o = Renderer(scene, seeds=seeds), orc.OracleRenderer(scene, seeds=seeds)
    r.set_debug_flags(FLAGS)
    r.run_samples(1); o.run_sample()
    r.run_samples(2); o.run_sample(); o.run_sample()
    agg = r.export_aggregators()
    ok = all(agg[f].tobytes() == o.weight_aggregators[f].tobytes() for f in ("weights", "total_contribution", "contrib_weight_sum"))
    ok &= np.array_equal(r.get_random_buffer(), o.rand_buffer) and r.counters()["rays"] == o.rays_traced
    ok &= r.read_accumulators()[3].tobytes() == o.unidirectional_image_buffer.tobytes()
    ok &= bool(np.allclose(r.read_accumulators()[0], o.summed_image, rtol=5e-5, atol=1e-8))
    print(name, W, H, "flags %#x" % FLAGS, "OK" if ok else "MISMATCH", flush=True)
    good &= ok
    r.close()
sys.exit(0 if good else 1)
