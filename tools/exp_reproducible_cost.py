"""What the reproducible light image (cl2_set_reproducible: records + radix sort + ordered gather instead of float atomics) costs:
    python tools/exp_reproducible_cost.py [scene=cornell] [K=1]
ms per sample of run_samples() with the switch off / on (alternating, twice), the serial resolve-stage time of each, and whether
two renders with the switch on give the same bytes."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer, stream_seeds
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = 1920, 1080
scene, desc = bench.build_scene(name, W, H)
N = max(2, {"cornell": 96, "glass": 48, "blob": 32, "interior": 16}.get(name, 32) // K)
print(desc, "K", K)
hashes = {0: [], 1: []}
for rep in range(2):
    for on in (0, 1):
        r = Renderer(scene, seeds=stream_seeds(W * H, K), streams=K)
        r.set_reproducible(on)
        r.tune(); r.reset_accumulators(); r.set_seeds(stream_seeds(W * H, K))
        r.run_samples(2); r.synchronize()
        t = time.perf_counter(); r.run_samples(N); r.synchronize(); dt = time.perf_counter() - t
        hashes[on].append(hashlib.sha256(r.packed_accumulators().tobytes()).hexdigest()[:12])
        r.reset_counters(); r.set_profiling(2); r.set_pipelining(0); r.run_samples(max(1, 4 // K))
        c = r.counters()
        print("reproducible %d  %8.3f ms/sample   serial resolve stage %.3f ms/sample   accumulators %s" %
              (on, dt / (N * K) * 1e3, c["ms_connect_resolve"] / (max(1, 4 // K) * K), hashes[on][-1]), flush=True)
        r.close()
print("two renders with the switch on agree byte for byte:", hashes[1][0] == hashes[1][1], "| with float atomics:", hashes[0][0] == hashes[0][1])
