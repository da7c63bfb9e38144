"""Sample streams sweep:  python tools/exp_streams.py <scene> [WxH] K1 K2 ...   (e.g. glass 1,2,4,8)
For each K: a renderer with K sample streams, tuner in the warm-up, then `total` samples of the frame (total / K passes)
timed; prints ms per SAMPLE, Grays/s and the tuned stage share.  Same total sample count for every K."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    name = sys.argv[1]
    W, H = 1920, 1080
    args = sys.argv[2:]
    if args and "x" in args[0]:
        W, H = (int(x) for x in args.pop(0).split("x"))
    ks = [int(k) for k in args] or [1, 2, 4, 8]
    total = int(os.environ.get("EXP_TOTAL", "96"))
    from clive2_amd.renderer import Renderer
    scene, desc = bench.build_scene(name, W, H)
    print(desc, f"{W}x{H}", flush=True)
    for K in ks:
        r = Renderer(scene, streams=K)
        tuned = r.tune()
        r.run_samples(2)
        r.reset_counters(); r.synchronize()
        n = max(1, total // K)
        t0 = time.perf_counter()
        r.run_samples(n)
        r.synchronize()
        dt = time.perf_counter() - t0
        c = r.counters()
        org = r.organisation()
        print(f"K {K}: {dt / (n * K) * 1e3:8.3f} ms per sample  {c['rays'] / dt / 1e9:6.3f} Grays/s  passes {n}  share {org['paths_share']}  "
              f"tuner passes {tuned}  stages {org['pipeline_stages']}", flush=True)
        r.close()


if __name__ == "__main__":
    main()
