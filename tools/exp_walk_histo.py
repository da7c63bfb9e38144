"""Pass statistics of the exact 4-wide walk (instrumentation build, -DCL2_WALK_HISTO):
    hipcc <flags of clive2_amd/_native.py> -DCL2_WALK_HISTO clive2_amd/csrc/*.hip -o gpurun_out/lib_histo.so
    python tools/exp_walk_histo.py gpurun_out/lib_histo.so <scene> [streams]
Prints, for the connection-ray launch and for the per-level subpath launches of one 1080p pass: the distribution of lanes
that visit a node / test triangles (first, second pair round) / do nothing per pass, and of triangles per entered leaf."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import clive2_amd._native as n
n.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
from clive2_amd.renderer import Renderer


def show(tag, h):
    names = ["node lanes", "tri lanes (1st round)", "tri lanes (2nd round)", "lanes with nothing to do"]
    passes = h[0].sum()
    print(f"== {tag}: {passes} passes")
    for k in range(4):
        tot = h[k].sum()
        mean = (h[k] * np.arange(65)).sum() / max(tot, 1)
        q = [int((h[k][lo:hi]).sum() * 100 / max(tot, 1)) for lo, hi in ((0, 1), (1, 9), (9, 17), (17, 33), (33, 65))]
        print(f"  {names[k]:28s} mean {mean:5.1f}   share of passes with 0 / 1-8 / 9-16 / 17-32 / 33-64 lanes: {q}")
    leaf = h[4][:17]
    print("  triangles per entered leaf:", {i: int(v) for i, v in enumerate(leaf) if v}, "mean %.2f" % ((leaf * np.arange(17)).sum() / max(leaf.sum(), 1)))
    deep = h[5]
    print("  deepest stack of a wave's lanes per pass: mean %.2f, share of passes with > 5 entries (no room to speculate): %.1f %%" %
          ((deep * np.arange(65)).sum() / max(deep.sum(), 1), 100.0 * deep[6:].sum() / max(deep.sum(), 1)))


def main():
    scene, desc = bench.build_scene(sys.argv[2], 1920, 1080)
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    r = Renderer(scene, streams=K)
    L = r._L
    buf = np.zeros((6, 65), np.uint64)                        # g_walk_histo[6][65]: the sixth row is the deepest stack per pass
    get = lambda: (L.cl2_walk_histo(r._h, buf.ctypes.data_as(C.c_void_p)), buf.copy())[1]
    r.set_debug_flags(8 | (int(sys.argv[4], 0) if len(sys.argv) > 4 else 0))      # (+ e.g. 0x2000: without the speculative expansion)
    r.set_traversal_mode(5)
    r.make_light_rays(); r.make_camera_rays()
    get()
    r.trace_light_rays(); r.trace_camera_rays()
    show(f"{sys.argv[2]} subpath launches (12 per-level launches)", get())
    r.join_paths()
    show(f"{sys.argv[2]} connection launch", get())


if __name__ == "__main__":
    main()
