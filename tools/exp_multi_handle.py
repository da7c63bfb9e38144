"""Does overlapping INDEPENDENT subpath chains hide the launch tails?  (GPU box only)
N renderers on one GPU, each rendering a 1/N-height strip-sized frame of the same scene from its own host thread,
against one renderer on the full frame: aggregate Grays/s.  The per-pixel chain light L0..L5 -> camera C0..C5 -> next
sample is strictly serial (one RNG stream per pixel), but different pixels are independent: if N concurrent chains beat
one, the subpath stage should be cut into pixel groups on separate streams.
    python tools/exp_multi_handle.py glass 1 2 4 [mode] [pipelining]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from clive2_amd.renderer import Renderer, make_seeds


def main():
    name = sys.argv[1]
    counts = [int(x) for x in sys.argv[2:] if x.isdigit() and int(x) <= 16][:4] or [1, 2, 4]
    mode = int(os.environ.get("MODE", "0")); pipe = int(os.environ.get("PIPE", "-1")); n = int(os.environ.get("SAMPLES", "48"))
    W, H = 1920, 1080
    for k in counts:
        scene, desc = bench.build_scene(name, W, H // k)
        rs = [Renderer(scene, seeds=make_seeds(W * (H // k), seed=100 + i)) for i in range(k)]
        for r in rs:
            r.set_traversal_mode(mode); r.set_pipelining(pipe)
            r.tune(); r.run_samples(4); r.reset_counters()
        bar = threading.Barrier(k + 1)
        def work(r):
            bar.wait(); r.run_samples(n); r.synchronize()
        th = [threading.Thread(target=work, args=(r,)) for r in rs]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        rays = sum(r.counters()["rays"] for r in rs)
        org = rs[0].organisation()
        print(f"{name}: {k} renderer(s) of {W}x{H // k}: {dt / n * 1e3:8.3f} ms per sample of the whole frame, {rays / dt / 1e9:6.3f} Grays/s  (share {org['paths_share']}, mode {mode}, pipe {pipe})", flush=True)
        for r in rs: r.close()


if __name__ == "__main__":
    main()
