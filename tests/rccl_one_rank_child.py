"""Child process of test_rccl_reduce_path_on_one_rank: a one-rank nccl (= RCCL) group on cuda:0, the
accumulators handed to all_reduce as a device buffer and written back.  Prints `STEP <name>` lines
so that the parent can tell where a hang happened."""
import os
import socket
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def step(name):
    print("STEP", name, flush=True)


def main():
    import torch
    import torch.distributed as dist
    from clive2_amd.renderer import Renderer, make_seeds
    from clive2_amd.scene import create_scene_from_preset
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    scene = create_scene_from_preset("empty", pixel_width=64, pixel_height=48)
    r = Renderer(scene, seeds=make_seeds(64 * 48))
    r.run_samples(2)
    before = r.packed_accumulators().copy()
    step("rendered")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    step("group-up")
    r.reduce_accumulators(always=True)
    assert r.packed_accumulators().tobytes() == before.tobytes(), "sum over one rank changed the accumulators"
    step("handover-ok")
    r.run_samples(1)                                  # the renderer keeps working after the hand-over
    assert np.isfinite(r.packed_accumulators()).all()
    step("render-after-ok")
    r.close()
    dist.destroy_process_group()
    step("group-down")


if __name__ == "__main__":
    main()
