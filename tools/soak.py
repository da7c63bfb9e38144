"""Long runs: 1500 spp at 1080p and 64 spp at 4K on the Cornell box, 300 spp of the 5k-triangle glass
scene in the persistent organisation (pipeline tuner included), many short calls in a row; finite
accumulators, exact sample counts, stable rate.   python tools/soak.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, clive2_amd as c2
from clive2_amd import meshes
from clive2_amd.load import get_materials
from clive2_amd.renderer import Renderer, make_seeds


def report(tag, r, n, dt):
    img, wts, cnt, uni = r.read_accumulators()
    c = r.counters()
    print(f"{tag} x{n}: {dt:.2f}s  {c['rays']/dt/1e9:.2f} Grays/s  finite={np.isfinite(img).all()} nan_w={np.isnan(wts).sum()} "
          f"cnt_ok={(cnt==n).all()} mean={img.mean()/n:.5f} rad_mean={r.radiance.mean():.5f} max={r.radiance.max():.3f}", flush=True)


for (w, h, n) in ((1920, 1080, 1500), (3840, 2160, 64)):
    s = c2.create_scene_from_preset("empty", w, h)
    r = Renderer(s, seeds=make_seeds(w * h))
    t = time.time(); r.run_samples(n); dt = time.time() - t
    report(f"cornell {w}x{h}", r, n, dt)
    r.close()

mats = get_materials(); mats["alpha"][5] = 0.1
s = c2.create_scene(1920, 1080, np.array([0, 1.5, 6]), np.array([0, 0, -1]), materials=mats,
                    file_specs=[dict(mesh=meshes.icosphere(4, radius=2.0, center=(0.0, 1.0, 0.0)), material=5)])
r = Renderer(s, seeds=make_seeds(1920 * 1080))
t = time.time(); r.run_samples(300); dt = time.time() - t
report("glass 1920x1080", r, 300, dt)
r.close()

s = c2.create_scene_from_preset("empty", 640, 360)
r = Renderer(s, seeds=make_seeds(640 * 360))
t = time.time()
for k in range(400):                       # many short calls: 1, 2, 3, 5 samples (buffer-set rotation across calls)
    r.run_samples((1, 2, 3, 5)[k % 4])
dt = time.time() - t
report("cornell 640x360, 400 short calls", r, 1100, dt)
r.close()

# round 4: sample streams -- 1024 spp of the 82k-triangle scene (BASELINE config 4's sample count) and 256 spp of the 1M-triangle
# scene with 8 streams, then stream-count changes on a live handle (per-pixel state re-allocated each time, accumulators kept)
import bench
for name, spp in (("blob", 1024), ("interior", 256)):
    s, desc = bench.build_scene(name, 1920, 1080)
    r = Renderer(s, streams=8)
    t = time.time(); r.run_samples(spp // 8); dt = time.time() - t
    report(f"{name} 1920x1080, 8 streams", r, spp, dt)
    r.close()
s = c2.create_scene_from_preset("empty", 640, 360)
r = Renderer(s)
total, t = 0, time.time()
for k in range(60):
    K = (1, 3, 8, 2)[k % 4]
    r.set_sample_streams(K)
    r.set_seeds(np.stack([make_seeds(640 * 360, seed=k, rank=j) for j in range(K)]))
    r.run_samples(3)
    total += 3 * K
report("cornell 640x360, 60 stream-count changes", r, total, time.time() - t)
r.close()
