"""N > 1 path on CPU: world_size-2 gloo process group.  Each rank integrates its own samples
(own seed buffer) -- here with the oracle standing in for the GPU -- and the packed accumulators
are summed with the product's all-reduce helper; the result must equal the single-process sum and
the radiance of the combined run (accumulators are pure sums)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 24, 16, 3


def _packed_from_oracle(o):
    B = W * H
    a = np.zeros((8, B), np.float32)
    a[0:3] = o.summed_image.reshape(B, 3).T
    a[3] = o.summed_sample_weights.reshape(B)
    a[4:7] = o.unidirectional_image_buffer.reshape(B, 3).T
    a[7] = o.summed_sample_counts.reshape(B)
    return a.reshape(-1)


def _render_rank(rank):
    import clive2_amd as c2
    from clive2_amd.renderer import make_seeds
    from oracle import oracle as orc
    scene = c2.create_scene_from_preset("empty", W, H)
    o = orc.OracleRenderer(scene, seeds=make_seeds(W * H, rank=rank))
    for _ in range(SPP):
        o.run_sample()
    return _packed_from_oracle(o)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    from clive2_amd.distributed import allreduce_packed_host, rank_info
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert rank_info() == (rank, rank, world)
    total = allreduce_packed_host(_render_rank(rank))
    np.save(os.path.join(out_dir, f"total_{rank}.npy"), total)
    dist.destroy_process_group()


def test_two_rank_sum_reduce(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = (np.load(tmp_path / f"total_{r}.npy") for r in (0, 1))
    assert np.array_equal(t0, t1)                                  # all ranks hold the same sum
    expect = _render_rank(0) + _render_rank(1)
    assert np.array_equal(t0, expect)                              # two-term float sums are exact either way
    from clive2_amd.distributed import radiance_from_packed
    rad = radiance_from_packed(t0, H, W)
    assert rad.shape == (H, W, 3) and np.isfinite(rad).all() and rad.mean() > 0
    assert (t0.reshape(8, -1)[7] == 2 * SPP).all()                 # sample counts add up


def test_sample_partition():
    from clive2_amd.distributed import samples_for_rank
    for total, world in ((1024, 8), (10, 4), (3, 8)):
        parts = [samples_for_rank(total, r, world) for r in range(world)]
        assert sum(parts) == total and max(parts) - min(parts) <= 1


def test_movie_frames_are_split_without_overlap():
    """Turntable frames are independent units (movie.py:29-55): every frame goes to exactly one rank."""
    from clive2_amd.movie import frames_for_rank
    for start, total, world in ((0, 120, 8), (7, 120, 8), (0, 5, 8), (3, 3, 2), (0, 1, 1)):
        parts = [frames_for_rank(start, total, r, world) for r in range(world)]
        flat = sorted(f for p in parts for f in p)
        assert flat == list(range(start, total))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---- RCCL bootstrap (id exchange only; no GPU, no RCCL, no torch) ----
def _bootstrap_worker(rank, world, path, out_dir, delay):
    sys.path.insert(0, ROOT)
    import time
    from clive2_amd.distributed import exchange_unique_id, finish_exchange
    time.sleep(delay[rank])
    made = []

    def make_id():                       # stands in for cl2_comm_get_unique_id: only rank 0 may call it
        made.append(1)
        return bytes(np.random.RandomState(os.getpid() % 2 ** 31).randint(0, 256, 128, dtype=np.uint8))

    uid = exchange_unique_id(rank, world, make_id, 128, path=path, timeout=30.0)
    assert (len(made) == 1) == (rank == 0)
    with open(os.path.join(out_dir, f"id_{rank}.bin"), "wb") as f:
        f.write(uid)
    # cl2_comm_init_rank would come here: it is a collective, so every rank holds the id when it returns
    while not all(os.path.exists(os.path.join(out_dir, f"id_{r}.bin")) for r in range(world)):
        time.sleep(0.01)
    finish_exchange(rank, path)


@pytest.mark.parametrize("delays", [(0.0, 0.3, 0.1), (0.4, 0.0, 0.0)])
def test_unique_id_bootstrap_across_processes(tmp_path, delays):
    """Rank 0 publishes the communicator id through a file (write + rename), the other ranks poll for it --
    whichever starts first -- and every rank ends up with the same 128 bytes; rank 0 removes the file."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "rdv" / "id")
    os.makedirs(os.path.dirname(path))
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, 3, path, str(tmp_path), delays)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ids = [open(tmp_path / f"id_{r}.bin", "rb").read() for r in range(3)]
    assert len(ids[0]) == 128 and ids[0] == ids[1] == ids[2]
    assert not os.path.exists(path) and os.listdir(os.path.dirname(path)) == []      # nothing left behind


def test_unique_id_bootstrap_edge_cases(tmp_path, monkeypatch):
    from clive2_amd import distributed as d
    # one rank: no file at all
    assert d.exchange_unique_id(0, 1, lambda: b"x" * 128, 128, path=str(tmp_path / "none")) == b"x" * 128
    assert not os.path.exists(tmp_path / "none")
    # no rank 0: the others give up with an error instead of waiting for ever
    with pytest.raises(TimeoutError):
        d.exchange_unique_id(1, 2, None, 128, path=str(tmp_path / "never"), timeout=0.2)
    # a file left behind by a crashed job long ago is not mistaken for this job's id
    stale = tmp_path / "stale"
    stale.write_bytes(b"s" * 128)
    os.utime(stale, (1.0, 1.0))
    with pytest.raises(TimeoutError):
        d.exchange_unique_id(1, 2, None, 128, path=str(stale), timeout=0.2)
    # a wrong-sized id is refused by rank 0
    with pytest.raises(ValueError):
        d.exchange_unique_id(0, 2, lambda: b"short", 128, path=str(tmp_path / "bad"))
    # the default path separates launches: it depends on the port, the run id and the launcher's pid
    monkeypatch.delenv("CLIVE2_RENDEZVOUS_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29511")
    a = d.rendezvous_path()
    monkeypatch.setenv("MASTER_PORT", "29512")
    assert d.rendezvous_path() != a and str(os.getppid()) in a
    monkeypatch.setenv("CLIVE2_RENDEZVOUS_FILE", "/tmp/explicit")
    assert d.rendezvous_path() == "/tmp/explicit"
