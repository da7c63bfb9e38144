import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import clive2_amd as c2
from clive2_amd.renderer import Renderer, make_seeds
scene = c2.create_scene_from_preset("empty", 1920, 1080)
for flags in (0, 1 << 7, 0, 1 << 7):
    r = Renderer(scene, seeds=make_seeds(1920*1080))
    r.set_debug_flags(flags)
    r.run_samples(32); r.synchronize()
    t = time.perf_counter(); r.run_samples(128); r.synchronize(); dt = time.perf_counter() - t
    print(flags, r.organisation()["pruned_records"], "ms/sample %.3f" % (dt / 128 * 1e3), flush=True)
    r.close()
