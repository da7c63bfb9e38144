"""Single run_sample() calls (the reference's loop, src/render.py:31-37): per-stage HIP-event times per sample, one sample stream.
    python tools/exp_serial_single.py <scene> [n=8] [WxH]      (CL2_LIB=build/lib_x.so picks a build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clive2_amd._native as _n
if os.environ.get("CL2_LIB"):
    _n.LIB_PATH = os.path.abspath(os.environ["CL2_LIB"])
import bench
from clive2_amd.renderer import Renderer
W, H = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1920x1080").split("x"))
scene, desc = bench.build_scene(sys.argv[1], W, H)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = Renderer(scene)
for _ in range(2):
    r.run_sample()
r.reset_counters(); r.set_profiling(2)
t0 = time.perf_counter()
for _ in range(n):
    r.run_sample()
r.synchronize()
dt = time.perf_counter() - t0
c = r.counters()
print(desc, "%dx%d" % (W, H))
print("  single run_sample(): %.3f ms per call (host clock);" % (dt / n * 1e3),
      {k[3:]: round(c[k] / n, 3) for k in c if k.startswith("ms_") and c[k] > 0},
      "subpath launches per sample %.1f" % (c["launches_traverse_paths"] / n))
