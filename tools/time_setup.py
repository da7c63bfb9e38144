"""Where the set-up time of a big scene goes (GPU box):  python tools/time_setup.py [interior|blob] [native|gpu|numpy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clive2_amd as c2
from clive2_amd import meshes, bvh, load
from clive2_amd.load import get_materials
from clive2_amd.renderer import Renderer


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "interior"
    builder = sys.argv[2] if len(sys.argv) > 2 else "native"
    t = [time.perf_counter()]
    def lap(what):
        t.append(time.perf_counter()); print(f"{what:42s} {t[-1] - t[-2]:7.3f} s", flush=True)
    if name == "interior":
        specs = [dict(mesh=(v, f), material=m) for v, f, m in meshes.interior_grid()]
    else:
        specs = [dict(mesh=meshes.noisy_blob(subdiv=6), material=5)]
    lap("mesh generation")
    from clive2_amd.camera import Camera
    cam = Camera(center=np.array([0, 1.5, 6]), direction=np.array([0, 0, -1]), pixel_width=1920, pixel_height=1080, phys_width=16 / 9, phys_height=1)
    soups = [bvh.FastTreeBox.from_triangle_objects(load.camera_geometry(cam) + load.triangles_for_box())]
    for s in specs:
        v, f = s["mesh"]
        soups.append(load.fast_load(np.asarray(v), np.asarray(f), material=s["material"]))
    lap("fast_load (smooth normals, bounds)")
    soup = bvh.FastTreeBox.concat(soups)
    lap("concat")
    root = bvh.construct_BVH(soup, builder=builder)
    lap(f"construct_BVH ({builder})")
    boxes, tris = bvh.np_flatten_bvh(root)
    lap("np_flatten_bvh (fill Triangle records)")
    t0 = time.perf_counter()
    scene = c2.create_scene(1920, 1080, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=get_materials(), bvh_builder=builder)
    t.append(time.perf_counter()); print(f"{'create_scene, all of the above':42s} {t[-1] - t0:7.3f} s", flush=True)
    r = Renderer(scene)
    lap("Renderer(): validate, repack, upload")
    r.run_samples(8)
    c = r.counters()
    lap("8 samples")
    r.set_counting(True); r.run_samples(1); c = r.counters()
    print(f"{len(scene.triangles)} tris / {len(scene.boxes)} boxes; node tests per ray {c['box_tests'] / c['counted_rays']:.1f}, triangle tests {c['tri_tests'] / c['counted_rays']:.1f}")


if __name__ == "__main__":
    main()
