"""Experiment matrix for the whole-subpath persistent launch (traversal mode 4): register budget (waves per SIMD),
bounce batching (lanes gathered / steps waited), against the per-level organisation (mode 2).  GPU box only.
    python tools/exp_whole.py glass 1920 1080"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from clive2_amd.renderer import Renderer, make_seeds


def run(scene, W, H, mode, flags, pipelining, n=24, gather=(0, 0)):
    r = Renderer(scene, seeds=make_seeds(W * H))
    r.set_traversal_mode(mode); r.set_debug_flags(flags); r.set_subpath_gather(*gather); r.set_pipelining(pipelining)
    r.run_samples(4)
    r.reset_counters()
    t0 = time.perf_counter()
    r.run_samples(n)
    dt = time.perf_counter() - t0
    rays = r.counters()["rays"]
    uni = r.read_accumulators()[3].copy()
    r.close()
    return dt / n * 1e3, rays / dt / 1e9, uni


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "glass"
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    scene, desc = bench.build_scene(name, W, H)
    print(desc, flush=True)
    ms, gr, ref = run(scene, W, H, 2, 0, 0)
    print(f"mode 2 serial            {ms:8.3f} ms  {gr:6.3f} Grays/s", flush=True)
    ms, gr, u = run(scene, W, H, 2, 0, 1)
    print(f"mode 2 pipelined         {ms:8.3f} ms  {gr:6.3f} Grays/s", flush=True)
    for wps in (0,):
        for lanes, wait in ((8, 16), (16, 24), (24, 32), (32, 48), (48, 64)):
            flags = 0
            ms, gr, u = run(scene, W, H, 4, flags, 0, gather=(lanes, wait))
            same = u.tobytes() == ref.tobytes()
            ms1, gr1, _ = run(scene, W, H, 4, flags, 1, gather=(lanes, wait))
            print(f"mode 4 wps {wps if wps < 7 else 8} lanes {lanes:2d} wait {wait:2d}: serial {ms:8.3f} ms {gr:6.3f} Grays/s | pipelined {ms1:8.3f} ms {gr1:6.3f} Grays/s"
                  f"  {'same' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
