"""Command-line render: the flag surface of the reference's `src/render.py:13-19`, writing a PNG
instead of opening a cv2 window (display code is out of scope).

    python -m clive2_amd.render --scene empty --width 1280 --height 720 --samples 64 --out cornell.png

Several GPUs: `python -m torch.distributed.run --nproc-per-node N -m clive2_amd.render ...` -- every
rank renders its share of the samples of the same frame with its own seeds, ONE sum all-reduce of the
accumulators (RCCL) follows, rank 0 writes the picture.
"""
import argparse
import time

import numpy as np

from .distributed import rank_info, samples_for_rank
from .renderer import Renderer, RendererError, make_seeds
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

    rank, local_rank, world = rank_info()
    if world > 1:
        import torch
        import torch.distributed as dist
        local_rank %= max(torch.cuda.device_count(), 1)       # one visible GPU per rank: it is device 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    scene = create_scene_from_preset(args.scene, pixel_width=args.width, pixel_height=args.height)
    device = local_rank if world > 1 else args.device
    renderer = Renderer(scene, seeds=make_seeds(args.width * args.height, rank=rank), device=device)
    t0 = time.time()
    try:
        renderer.run_samples(samples_for_rank(args.samples, rank, world))
    except (KeyboardInterrupt, RendererError):
        if not args.save_on_quit:
            raise
    renderer.reduce_accumulators()
    dt = time.time() - t0
    rays = renderer.counters()["rays"]
    print(f"[rank {rank}] rendering took {dt:.2f} seconds ({renderer.samples} samples, {rays / max(dt, 1e-9) / 1e6:.0f} Mrays/s)")
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return 0
    image = renderer.image                      # tone-mapped uint8, BGR, row 0 = bottom of the film
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(image[::-1, :, ::-1])).save(args.out)
    except ImportError:
        np.save(args.out + ".npy", image)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
