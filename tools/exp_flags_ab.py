"""Same-box A/B of debug-flag settings (include/clive2_amd.h) on the 1080p Cornell box:  python tools/exp_flags_ab.py 0 0x800 ...
One process, three alternating rounds; ms per sample of the default (pipelined) organisation, the serial per-stage breakdown and a
hash of the unidirectional accumulators (every setting must give the same one)."""
import sys, time, os, hashlib
sys.path.insert(0, os.getcwd())
import bench
from clive2_amd.renderer import Renderer, make_seeds
scene, desc = bench.build_scene("cornell", 1920, 1080)
flags = [int(x, 0) for x in sys.argv[1:]] or [0, 1 << 11]
for rep in range(3):
    for f in flags:
        r = Renderer(scene, seeds=make_seeds(1920 * 1080))
        r.set_debug_flags(f)
        r.tune(); r.run_samples(8); r.synchronize()
        N = 192
        t = time.perf_counter(); r.run_samples(N); r.synchronize(); dt = time.perf_counter() - t
        h = hashlib.sha256(r.read_accumulators()[3].tobytes()).hexdigest()[:10]
        r.reset_counters(); r.set_profiling(2); r.set_pipelining(0); r.run_samples(8)
        c = r.counters()
        st = {k[3:]: round(c[k] / 8, 3) for k in c if k.startswith("ms_") and c[k] > 0}
        print("flags %#x  %8.3f ms/sample  uni %s  serial %s" % (f, dt / N * 1e3, h, st), flush=True)
        r.close()
