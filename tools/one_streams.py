"""One scene, K sample streams, a few pipelined passes after the tuner (for rocprofv3 passes):
    python3 tools/one_streams.py <scene> <K> [passes] [WxH]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer
W, H = 1920, 1080
if len(sys.argv) > 4:
    W, H = (int(x) for x in sys.argv[4].split("x"))
scene, desc = bench.build_scene(sys.argv[1], W, H)
K = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
r = Renderer(scene, streams=K)
if os.environ.get("ONE_TUNE", "1") == "1":
    r.tune()
r.synchronize()
t0 = time.perf_counter()
r.run_samples(n)
r.synchronize()
dt = time.perf_counter() - t0
print("done", desc, "K", K, "ms per sample", round(dt / (n * K) * 1e3, 3), "share", r.organisation()["paths_share"], "rays", r.counters()["rays"])
