"""What the HIP-event timers of cl2_set_profiling cost inside a timed region: 1080p Cornell box, run_samples(n) repeated, levels 0 / 1 / 2.
    python tools/exp_profiling_overhead.py [n=20] [repeats=12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scene, _ = bench.build_scene("cornell", 1920, 1080)
r = Renderer(scene)
r.tune(); r.run_samples(8)
for rnd in range(3):
    for level in (0, 1, 2):
        r.set_profiling(level)
        ts = []
        for _ in range(reps):
            r.synchronize(); t0 = time.perf_counter(); r.run_samples(n); r.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
        r.counters()
        ts.sort()
        print("profiling %d: median %.4f ms per sample, best %.4f (run_samples(%d) x %d)" % (level, ts[len(ts) // 2], ts[0], n, reps), flush=True)
