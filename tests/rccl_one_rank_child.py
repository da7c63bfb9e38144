"""Child process of test_rccl_reduce_path_on_one_rank: a one-rank RCCL communicator on device 0 created
through the C ABI (cl2_comm_get_unique_id / cl2_comm_init_rank), the accumulators reduced in place by
cl2_reduce_accumulators, the host-value all-reduce, teardown.  Prints `STEP <name>` lines so that the
parent can tell where a hang happened.  torch is never imported."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def step(name):
    print("STEP", name, flush=True)


def main():
    from clive2_amd.renderer import Renderer, make_seeds
    from clive2_amd.scene import create_scene_from_preset
    from clive2_amd.distributed import join_communicator
    scene = create_scene_from_preset("empty", pixel_width=64, pixel_height=48)
    r = Renderer(scene, seeds=make_seeds(64 * 48))
    r.run_samples(2)
    before = r.packed_accumulators().copy()
    step("rendered")
    join_communicator(r, 0, 1)
    step("comm-up")
    r.reduce_accumulators()
    assert r.packed_accumulators().tobytes() == before.tobytes(), "sum over one rank changed the accumulators"
    step("handover-ok")
    assert r.allreduce_host([1.5, -2.0, 7.0], op="sum") == [1.5, -2.0, 7.0]
    assert r.allreduce_host([1.5, -2.0], op="max") == [1.5, -2.0]
    step("host-allreduce-ok")
    r.run_samples(3)                                  # the renderer keeps working after the collective (pipelined samples)
    after = r.packed_accumulators().copy()
    assert np.isfinite(after).all() and (after.reshape(8, -1)[7] == 5).all()
    step("render-after-ok")
    r.reduce_accumulators()
    assert r.packed_accumulators().tobytes() == after.tobytes()
    step("second-reduce-ok")
    r.comm_destroy()
    step("comm-down")
    r.close()
    step("closed")
    print("MODULES", " ".join(sorted(m for m in sys.modules if m.split(".")[0] in ("torch",))))


if __name__ == "__main__":
    main()
