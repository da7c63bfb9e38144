import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from clive2_amd.renderer import Renderer
for scene_name in ("glass", "interior"):
    scene, desc = bench.build_scene(scene_name, 1920, 1080)
    for K in (8,):
        for mode in (0, 4):
            r = Renderer(scene, streams=K)
            r.set_traversal_mode(mode)
            r.tune(); r.run_samples(1); r.synchronize()
            n = 6
            t0 = time.perf_counter(); r.run_samples(n); r.synchronize(); dt = time.perf_counter() - t0
            print(scene_name, "K", K, "mode", mode, "ms per sample %.3f" % (dt / (n * K) * 1e3), "share", r.organisation()["paths_share"], flush=True)
            r.close()
