"""Serial-order stage breakdown (HIP events around every launch, each launch alone on the machine) with K sample streams:
    python tools/exp_stage_breakdown.py <scene> <K> [passes]
ms per SAMPLE per stage, launches and rays of the two traversal stages, and the pipelined figure beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clive2_amd._native as _n
if os.environ.get("CL2_LIB"):
    _n.LIB_PATH = os.path.abspath(os.environ["CL2_LIB"])      # A/B of builds: CL2_LIB=build/lib_x.so
import bench
from clive2_amd.renderer import Renderer
scene, desc = bench.build_scene(sys.argv[1], 1920, 1080)
K = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r = Renderer(scene, streams=K)
r.set_debug_flags(8)                     # the wide walk for the per-level subpath launches in the serial order too
if K * 1920 * 1080 <= (1 << 22):
    r.set_traversal_mode(5)              # per-level launches (the automatic choice would take whole subpaths in the serial order)
r.set_pipelining(0); r.run_samples(2); r.reset_counters(); r.set_profiling(2)
r.run_samples(n)
c = r.counters()
S = n * K
print(desc, "K", K)
print("  serial ms per sample:", {k[3:]: round(c[k] / S, 3) for k in c if k.startswith("ms_") and c[k] > 0}, "sum", round(sum(c[k] for k in c if k.startswith("ms_")) / S, 3))
print("  rays per sample: subpath %.1f M in %d launches, connection %.1f M" % (c["rays_traverse_paths"] / S / 1e6, c["launches_traverse_paths"] / n, c["rays_traverse_conn"] / S / 1e6))
print("  Grays/s of the traversal launches alone: subpath %.2f, connection %.2f" % (c["rays_traverse_paths"] / c["ms_traverse_paths"] / 1e6, c["rays_traverse_conn"] / c["ms_traverse_conn"] / 1e6))
r.set_profiling(0); r.set_pipelining(-1); r.set_debug_flags(0); r.set_traversal_mode(0)
r.tune(); r.synchronize()
t0 = time.perf_counter(); r.run_samples(max(2, 48 // K)); r.synchronize(); dt = time.perf_counter() - t0
print("  pipelined: %.3f ms per sample, share %d" % (dt / (max(2, 48 // K) * K) * 1e3, r.organisation()["paths_share"]))
