"""Command-line render: the flag surface of the reference's `src/render.py:13-19`, writing a PNG
instead of opening a cv2 window (display code is out of scope).

    python -m clive2_amd.render --scene empty --width 1280 --height 720 --samples 64 --out cornell.png
"""
import argparse
import time

import numpy as np

from .renderer import Renderer, RendererError
from .scene import create_scene_from_preset


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--samples", type=int, default=15)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--save-on-quit", action="store_true")
    ap.add_argument("--scene", type=str, default="empty")
    ap.add_argument("--out", type=str, default="render.png")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    scene = create_scene_from_preset(args.scene, pixel_width=args.width, pixel_height=args.height)
    renderer = Renderer(scene, device=args.device)
    t0 = time.time()
    try:
        renderer.run_samples(args.samples)
    except (KeyboardInterrupt, RendererError):
        if not args.save_on_quit:
            raise
    dt = time.time() - t0
    rays = renderer.counters()["rays"]
    print(f"Rendering took {dt:.2f} seconds ({renderer.samples} samples, {rays / max(dt, 1e-9) / 1e6:.0f} Mrays/s)")
    image = renderer.image                      # tone-mapped uint8, BGR, row 0 = bottom of the film
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(image[::-1, :, ::-1])).save(args.out)
    except ImportError:
        np.save(args.out + ".npy", image)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
