"""Turntable movie: the flag surface of the reference's `src/movie.py:13-21`, one PNG per frame
instead of a cv2 window.  Every frame is its own scene (the camera quad is part of the geometry,
`load.py:261-271`) and its own Renderer, exactly as in the reference's frame loop (`movie.py:29-55`).

Frames are independent units: with one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE in the
environment, e.g. under `python -m torch.distributed.run --nproc-per-node N -m clive2_amd.movie ...`)
rank r renders frames r, r+N, r+2N, ... on its own GPU with no collective.

    python -m clive2_amd.movie --scene empty --width 1280 --height 720 --samples 15 --movie-frames 120
"""
import argparse
import os
import shutil
import time

import numpy as np

from .distributed import rank_info
from .renderer import Renderer, RendererError
from .scene import create_scene_from_preset_with_params


def frames_for_rank(start_frame, total_frames, rank, world):
    """Round-robin split of [start_frame, total_frames) -- neighbouring frames cost about the same."""
    return list(range(start_frame + rank, total_frames, world))


def save_frame(path, image):
    """`image` is the Renderer's tone-mapped uint8 BGR picture; row 0 is the TOP of the picture (the film
    sits behind the pinhole), exactly what the reference shows with cv2 (movie.py:45-47): BGR -> RGB only."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(image[:, :, ::-1])).save(path)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--samples", type=int, default=15)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--scene", type=str, default="teapots")
    ap.add_argument("--movie-name", type=str, default="test-movie")
    ap.add_argument("--movie-frames", type=int, default=120)
    ap.add_argument("--start-frame", type=int, default=0)
    ap.add_argument("--out-root", type=str, default="../output")
    ap.add_argument("--host-tonemap", action="store_true",
                    help="tone-map every frame on the host with numpy (the reference's path, Renderer.image) instead of on the device")
    args = ap.parse_args(argv)

    rank, local_rank, world = rank_info()
    out_dir = os.path.join(args.out_root, args.movie_name)
    if world == 1 and args.start_frame == 0 and os.path.exists(out_dir):
        shutil.rmtree(out_dir)                       # movie.py:23-26: a fresh movie replaces the old one
                                                     # (with several ranks frames are only overwritten)
    os.makedirs(out_dir, exist_ok=True)

    for f in frames_for_rank(args.start_frame, args.movie_frames, rank, world):
        t0 = time.time()
        scene = create_scene_from_preset_with_params(args.scene, pixel_width=args.width, pixel_height=args.height,
                                                     frame_idx=f, total_frames=args.movie_frames)
        try:
            renderer = Renderer(scene, device=local_rank)
        except RendererError:
            if local_rank == 0:
                raise
            renderer = Renderer(scene, device=0)     # the launcher exposes one GPU per rank: it is device 0
        renderer.run_samples(args.samples)
        # a frame leaves the device tone-mapped (6 MB at 1080p; Renderer.image reads 66 MB of accumulators and maps them with numpy)
        save_frame(os.path.join(out_dir, f"frame_{f:04d}.png"), renderer.image if args.host_tonemap else renderer.tone_mapped("image"))
        del renderer, scene
        print(f"Frame {f} time: {time.time() - t0:.3f}", flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
