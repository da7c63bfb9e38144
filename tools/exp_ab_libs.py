"""Same-box A/B of two builds of the library: python tools/exp_ab_libs.py <lib_a.so> <lib_b.so> [scene=empty]
Each library runs in its own child process (one library per process), alternating, 1080p, ms per sample."""
import os, subprocess, sys

CHILD = r'''
import sys, time, os
sys.path.insert(0, os.getcwd())
import clive2_amd._native as n
n.LIB_PATH = sys.argv[1]
import clive2_amd as c2
from clive2_amd.renderer import Renderer, make_seeds
scene = c2.create_scene_from_preset(sys.argv[2], 1920, 1080)
r = Renderer(scene, seeds=make_seeds(1920 * 1080))
r.run_samples(32); r.synchronize()
t = time.perf_counter(); r.run_samples(192); r.synchronize(); dt = time.perf_counter() - t
print("%s  ms/sample %.3f" % (os.path.basename(sys.argv[1]), dt / 192 * 1e3), flush=True)
'''

if __name__ == "__main__":
    a, b = sys.argv[1], sys.argv[2]
    scene = sys.argv[3] if len(sys.argv) > 3 else "empty"
    for lib in (a, b, a, b, a, b):
        subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), scene], check=True)
