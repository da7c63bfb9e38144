"""Same-box A/B of debug-flag settings (include/clive2_amd.h) on a mesh scene with K sample streams:
    python tools/exp_mesh_flags_ab.py <scene> <K> <flags> [<flags> ...]      (CL2_LIB=build/lib_x.so picks a build)
One process, two alternating rounds; per setting: ms per SAMPLE of the default (tuned, pipelined) organisation, the serial
per-stage breakdown per sample with per-level launches (HIP events, each launch alone on the machine) and a hash of the
unidirectional accumulators (every setting must give the same one)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clive2_amd._native as _n
if os.environ.get("CL2_LIB"):
    _n.LIB_PATH = os.path.abspath(os.environ["CL2_LIB"])
import bench
from clive2_amd.renderer import Renderer, stream_seeds
W, H = (int(x) for x in os.environ.get("FRAME", "1920x1080").split("x"))
scene, desc = bench.build_scene(sys.argv[1], W, H)
K = int(sys.argv[2])
flags = [int(x, 0) for x in sys.argv[3:]] or [0]
N = {"glass": 64, "blob": 48, "interior": 24, "open": 64, "cornell": 128}.get(sys.argv[1], 32)
passes = max(2, N // K)
print(desc, "K", K, flush=True)
for rep in range(2):
    for f in flags:
        r = Renderer(scene, seeds=stream_seeds(W * H, K), streams=K)
        r.set_debug_flags(f)
        r.tune(); r.run_samples(2); r.synchronize()
        t = time.perf_counter(); r.run_samples(passes); r.synchronize(); dt = time.perf_counter() - t
        h = hashlib.sha256(r.read_accumulators()[3].tobytes()).hexdigest()[:10]
        share = r.organisation()["paths_share"]
        if K * W * H <= (1 << 22):
            r.set_traversal_mode(5)          # per-level launches in the serial order too
        nb = max(1, 8 // K)
        r.reset_counters(); r.set_profiling(2); r.set_pipelining(0); r.run_samples(nb)
        c = r.counters()
        st = {k[3:]: round(c[k] / (nb * K), 3) for k in c if k.startswith("ms_") and c[k] > 0}
        print("flags %#8x  %8.3f ms/sample  share %d  uni %s  serial/sample %s" % (f, dt / (passes * K) * 1e3, share, h, st), flush=True)
        r.close()
