"""Same-box A/B of builds of the library on a bench scene:  python tools/exp_ab_scene.py <scene> <lib.so> [<lib.so> ...]
Each library runs in its own child process, twice, alternating; 1080p; prints ms per sample of the default (pipelined)
organisation and the serial per-stage breakdown (HIP events, launch alone)."""
import os, subprocess, sys

CHILD = r'''
import sys, time, os
sys.path.insert(0, os.getcwd())
import clive2_amd._native as n
n.LIB_PATH = sys.argv[1]
import bench
from clive2_amd.renderer import Renderer, make_seeds
scene, desc = bench.build_scene(sys.argv[2], 1920, 1080)
r = Renderer(scene, seeds=make_seeds(1920 * 1080))
r.tune(); r.run_samples(8); r.synchronize()
N = int(sys.argv[3])
t = time.perf_counter(); r.run_samples(N); r.synchronize(); dt = time.perf_counter() - t
uni = r.read_accumulators()[3]
import hashlib
h = hashlib.sha256(uni.tobytes()).hexdigest()[:10]
r.reset_counters(); r.set_profiling(2); r.set_pipelining(0); r.run_samples(8)
c = r.counters()
st = {k[3:]: round(c[k] / 8, 3) for k in c if k.startswith("ms_") and c[k] > 0}
print("%-28s %8.3f ms/sample  share %d  uni %s  serial %s" % (os.path.basename(sys.argv[1]), dt / N * 1e3, r.organisation()["paths_share"], h, st), flush=True)
'''

if __name__ == "__main__":
    scene, libs = sys.argv[1], sys.argv[2:]
    n = {"cornell": 192, "glass": 64, "blob": 48, "interior": 32}.get(scene, 48)
    for rep in range(2):
        for lib in libs:
            subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), scene, str(n)], check=True)
