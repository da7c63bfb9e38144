"""Timing of launch organisations on one scene (GPU box only):  python tools/exp_modes.py glass [flags...]
Each extra argument is `mode:debug_flags:pipelining`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer, make_seeds


def run(scene, W, H, mode, flags, pipelining, n):
    r = Renderer(scene, seeds=make_seeds(W * H))
    r.set_traversal_mode(mode); r.set_debug_flags(flags); r.set_pipelining(pipelining)
    r.tune()
    r.run_samples(4)
    r.reset_counters()
    if pipelining == 0:
        r.set_profiling(2)
    t0 = time.perf_counter()
    r.run_samples(n)
    dt = time.perf_counter() - t0
    c = r.counters()
    rays = c["rays"]
    run.last = {k[3:]: round(c[k] / n, 3) for k in c if k.startswith("ms_") and c[k] > 0}
    uni = r.read_accumulators()[3].copy()
    org = r.organisation()
    r.close()
    return dt / n * 1e3, rays / dt / 1e9, uni, org


def main():
    name = sys.argv[1]
    W, H = 1920, 1080
    scene, desc = bench.build_scene(name, W, H)
    print(desc, flush=True)
    ref = None
    for spec in sys.argv[2:]:
        mode, flags, pipe = (int(x, 0) for x in spec.split(":"))
        n = 60 if pipe != 0 else 24
        ms, gr, uni, org = run(scene, W, H, mode, flags, pipe, n)
        ref = uni if ref is None else ref
        print(f"mode {mode} flags {flags:#x} pipe {pipe}: {ms:8.3f} ms {gr:6.3f} Grays/s  {'same' if uni.tobytes() == ref.tobytes() else 'DIFFERENT'}  "
              f"window {org['n_lds_records']} share {org['paths_share']} wide {org['wide_connections']} {run.last}", flush=True)


if __name__ == "__main__":
    main()
