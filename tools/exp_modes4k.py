"""As tools/exp_modes.py at 3840x2160:  python tools/exp_modes4k.py interior 2:0:0 0:0:0 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_modes, bench


def main():
    name = sys.argv[1]
    W, H = 3840, 2160
    scene, desc = bench.build_scene(name, W, H)
    print(desc, flush=True)
    for spec in sys.argv[2:]:
        mode, flags, pipe = (int(x, 0) for x in spec.split(":"))
        n = 50 if pipe != 0 else 10
        ms, gr, uni, org = exp_modes.run(scene, W, H, mode, flags, pipe, n)
        print(f"mode {mode} flags {flags:#x} pipe {pipe}: {ms:8.3f} ms {gr:6.3f} Grays/s share {org['paths_share']} {exp_modes.run.last}", flush=True)


if __name__ == "__main__":
    main()
