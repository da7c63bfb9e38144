"""One scene, one debug-flag setting, a few serial samples (for rocprofv3 passes):  python3 tools/one_flags.py <flags> <scene> [samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer, make_seeds
scene, desc = bench.build_scene(sys.argv[2], 1920, 1080)
r = Renderer(scene, seeds=make_seeds(1920 * 1080))
r.set_debug_flags(int(sys.argv[1], 0))
r.set_pipelining(0)
r.run_samples(int(sys.argv[3]) if len(sys.argv) > 3 else 3)
r.synchronize()
print("done", r.counters()["rays"])
