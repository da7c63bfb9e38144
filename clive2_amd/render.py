"""Command-line render: the flag surface of the reference's `src/render.py:13-19`, writing a PNG
instead of opening a cv2 window (display code is out of scope).

    python -m clive2_amd.render --scene empty --width 1280 --height 720 --samples 64 --out cornell.png

Several GPUs: start one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE in the environment (e.g.
`python -m torch.distributed.run --nproc-per-node N -m clive2_amd.render ...`; any spawner will do, torch
itself is not used) -- every rank renders its share of the samples of the same frame with its own seeds,
ONE in-place RCCL all-reduce of the accumulators follows, rank 0 writes the picture.
"""
import argparse
import time

import numpy as np

from . import _native
from .distributed import rank_info, samples_for_rank, join_communicator
from .renderer import Renderer, RendererError, stream_seeds
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
    ap.add_argument("--device-tonemap", action="store_true",
                    help="tone-map on the device (Renderer.tone_mapped) instead of on the host with numpy (Renderer.image, the reference's path)")
    ap.add_argument("--sample-streams", type=lambda v: v if v == "auto" else int(v), default=1,
                    help="K independent samples of the frame per pass (Renderer(streams=K): one seed buffer each, as K renderers "
                         "would hold; pays on mesh scenes, where it makes every launch K times larger); the samples are rounded up "
                         "to a multiple of K.  1 = the reference's single renderer; auto = by scene and frame size (Renderer.auto_streams)")
    ap.add_argument("--reproducible", action="store_true",
                    help="sum the light image in a fixed order (Renderer.set_reproducible) instead of with float atomics: two runs "
                         "then write the same bytes, as the reference's sort + gather chain does (renderer.py:212-250); slower")
    args = ap.parse_args(argv)

    rank, local_rank, world = rank_info()
    scene = create_scene_from_preset(args.scene, pixel_width=args.width, pixel_height=args.height)
    if world > 1:
        # one GPU per rank; a launcher that exposes a single GPU to each rank makes it device 0
        device = local_rank % max(_native.lib().cl2_device_count(), 1)
    else:
        device = args.device
    renderer = Renderer(scene, device=device, streams=args.sample_streams if args.sample_streams == "auto" else max(1, args.sample_streams))
    K = renderer.streams
    if args.reproducible:
        renderer.set_reproducible(True)
    # seed buffers of the job: stream k of rank r is buffer r * K + k
    renderer.set_seeds(stream_seeds(args.width * args.height, K, first_rank=rank * K))
    if world > 1:
        join_communicator(renderer, rank, world)
    t0 = time.time()
    failure = None
    try:
        renderer.run_samples(-(-samples_for_rank(args.samples, rank, world) // K))
    except (KeyboardInterrupt, RendererError) as e:
        failure = e
    if world > 1:
        # Agree on the outcome BEFORE the collective: a rank that failed must not leave its peers blocked in
        # the all-reduce, and a sum that lacks a rank's samples must not be mistaken for the picture.
        # (A rank that died outright cannot answer; its peers then fail in RCCL when the launcher kills the job.)
        try:
            bad = renderer.allreduce_host([1.0 if failure is not None else 0.0], op="max")[0] > 0.0
        except RendererError as e:
            bad, failure = True, failure or e
        if bad and not args.save_on_quit:
            renderer.close()
            if failure is not None:
                raise failure
            raise RendererError(f"[rank {rank}] another rank failed: the accumulators were not reduced")
        if bad:
            print(f"[rank {rank}] a rank stopped early: reducing what every rank has ({renderer.samples} samples here)")
        renderer.reduce_accumulators()
    elif failure is not None and not args.save_on_quit:
        raise failure
    dt = time.time() - t0
    rays = renderer.counters()["rays"]
    print(f"[rank {rank}] rendering took {dt:.2f} seconds ({renderer.samples} samples, {rays / max(dt, 1e-9) / 1e6:.0f} Mrays/s)")
    if rank != 0:
        renderer.close()
        return 0
    # Tone-mapped uint8, BGR.  The film sits BEHIND the pinhole, so the picture on it is already upright
    # when read row 0 first (row 0 looks up at the ceiling light): the reference hands `renderer.image`
    # to cv2 unflipped (render.py:35-37).  Only the channel order changes for a PNG (BGR -> RGB).
    image = renderer.tone_mapped("image") if args.device_tonemap else renderer.image
    renderer.close()
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(image[:, :, ::-1])).save(args.out)
    except ImportError:
        np.save(args.out + ".npy", image)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
