import sys, time; sys.path.insert(0, '.')
import numpy as np, clive2_amd as c2
from clive2_amd.renderer import Renderer, make_seeds
for (w, h, n) in ((1920, 1080, 1500), (3840, 2160, 64)):
    s = c2.create_scene_from_preset("empty", w, h)
    r = Renderer(s, seeds=make_seeds(w * h))
    t = time.time(); r.run_samples(n); dt = time.time() - t
    img, wts, cnt, uni = r.read_accumulators()
    c = r.counters()
    print(f"{w}x{h} x{n}: {dt:.2f}s  {c['rays']/dt/1e9:.2f} Grays/s  finite={np.isfinite(img).all()} nan_w={np.isnan(wts).sum()} "
          f"cnt_ok={(cnt==n).all()} mean={img.mean()/n:.5f} rad_mean={r.radiance.mean():.5f} max={r.radiance.max():.3f}", flush=True)
    r.close()
