"""N > 1 path on CPU: world_size-2 gloo process group.  Each rank integrates its own samples
(own seed buffer) -- here with the oracle standing in for the GPU -- and the packed accumulators
are summed over the group through host memory (the GPU path is `Renderer.reduce_accumulators`: RCCL inside
the library); the result must equal the single-process sum and the radiance of the combined run
(accumulators are pure sums)."""
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


def allreduce_packed_host(packed, group=None):
    """CPU-side rehearsal of the reduce (test code: torch.distributed / gloo stands in for RCCL): sums a packed
    accumulator array over the group through host memory.  Returns a new float32 numpy array."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(packed, dtype=np.float32).copy())
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    from clive2_amd.distributed import rank_info
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


def _build_wait_stub(tmp_path):
    import subprocess
    exe = str(tmp_path / "comm_wait_stub")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "comm_wait_stub.cpp"), "-o", exe], check=True)
    return exe


def test_collective_wait_has_a_deadline_and_aborts(tmp_path):
    """a rank that never joins a collective must not leave its peers blocked for ever.  The
    library waits for an enqueued collective with wait_collective() (csrc/comm_wait.hpp: stream + async-error
    polling under a deadline) and aborts the communicator on failure.  Two processes, CPU stand-ins for
    stream and communicator: rank 1 skips the collective, rank 0 must return an error within its deadline."""
    import subprocess
    import time
    exe = _build_wait_stub(tmp_path)
    d = tmp_path / "skip"; d.mkdir()
    t0 = time.monotonic()
    p0 = subprocess.Popen([exe, str(d), "0", "2", "1", "1.0"], stdout=subprocess.PIPE, text=True)
    p1 = subprocess.Popen([exe, str(d), "1", "2", "0", "1.0"], stdout=subprocess.PIPE, text=True)
    out1, _ = p1.communicate(timeout=30)
    out0, _ = p0.communicate(timeout=30)
    waited = time.monotonic() - t0
    assert p1.returncode == 0 and "skipped" in out1
    assert p0.returncode == 5 and "TIMEOUT" in out0            # -CL2_E_COMM, not a hang
    assert 0.9 < waited < 10.0
    assert (d / "aborted.0").exists()                           # the communicator was torn down (ncclCommAbort in the library)

    # positive control: both join (rank 1 late) -> both complete, nobody aborts
    d = tmp_path / "both"; d.mkdir()
    p0 = subprocess.Popen([exe, str(d), "0", "2", "1", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    p1 = subprocess.Popen([exe, str(d), "1", "2", "1", "20"], stdout=subprocess.PIPE, text=True)
    assert p0.wait(timeout=30) == 0 and p1.wait(timeout=30) == 0
    assert not list(d.glob("aborted.*"))

    # an asynchronous communicator error ends the wait at once, long before the deadline
    d = tmp_path / "async"; d.mkdir()
    (d / "async_error").write_text("x")
    t0 = time.monotonic()
    p0 = subprocess.run([exe, str(d), "0", "2", "1", "60"], stdout=subprocess.PIPE, text=True, timeout=30)
    assert p0.returncode == 5 and "ASYNC_ERROR detail 6" in p0.stdout and time.monotonic() - t0 < 5.0


def _run_bench(args, env_extra=None, timeout=120):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CLIVE2_RENDEZVOUS_FILE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus N` (the driver's command form) must start N rank processes
    itself.  `--dry-spawn` makes the ranks stop after the communicator-id hand-over (128 random bytes through the
    shared rendezvous file), so the whole launch path runs here without a GPU: N children, distinct ranks, one
    rendezvous path handed to all, the same id everywhere, ONE line on stdout, nothing left behind."""
    import json
    import tempfile
    before = set(os.listdir(tempfile.gettempdir()))
    p = _run_bench(["--gpus", "4", "--dry-spawn", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                        # rank 0's line only
    out = json.loads(lines[0])
    assert out["dry_spawn"] and out["n_gpus"] == 4 and out["ids_equal"]
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and [r["local_rank"] for r in ranks] == [0, 1, 2, 3]
    assert all(r["world"] == 4 for r in ranks)
    assert len({r["pid"] for r in ranks}) == 4                    # four processes ...
    assert len({r["ppid"] for r in ranks}) == 1                   # ... of one parent
    assert len({r["rendezvous"] for r in ranks}) == 1 and ranks[0]["rendezvous"] == ranks[0]["explicit_file"]
    left = [f for f in set(os.listdir(tempfile.gettempdir())) - before if f.startswith("clive2_bench_id_")]
    assert not left


def test_bench_spawn_propagates_a_rank_failure():
    """A rank that exits non-zero ends the job: the parent stops the other ranks (which would otherwise wait for
    the id / in a collective), exits non-zero itself and prints no result line."""
    import time
    t0 = time.monotonic()
    p = _run_bench(["--gpus", "3", "--dry-spawn"], env_extra={"CLIVE2_BENCH_DRY_FAIL_RANK": "0"})
    assert p.returncode == 3 and p.stdout.strip() == "" and "rank 0 exited with 3" in p.stderr
    assert time.monotonic() - t0 < 30.0                           # not the ranks' own 60 s id timeout
    p = _run_bench(["--gpus", "3", "--dry-spawn"], env_extra={"CLIVE2_BENCH_DRY_FAIL_RANK": "2"})
    assert p.returncode == 3 and p.stdout.strip() == ""


@pytest.mark.parametrize("how", ["SIGTERM", "SIGINT", "SIGKILL"])
def test_bench_spawn_leaves_no_rank_behind_when_the_parent_is_ended(how, tmp_path):
    """a parent that is interrupted or killed must not leave rank processes holding their GPUs.
    The ranks here never finish (CLIVE2_BENCH_DRY_HANG_DIR: they stand for ranks blocked in a collective).  SIGTERM /
    SIGINT: the parent's handlers fall into its clean-up (terminate, wait, kill), it exits 128 + signal and removes the
    rendezvous file.  SIGKILL: no handler runs; every child asked the kernel for a SIGTERM on its parent's death
    (PR_SET_PDEATHSIG) and is gone all the same."""
    import signal
    import subprocess
    import tempfile
    import time
    before = set(os.listdir(tempfile.gettempdir()))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CLIVE2_RENDEZVOUS_FILE")}
    env["CLIVE2_BENCH_DRY_HANG_DIR"] = str(tmp_path)
    parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-spawn"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        deadline = time.monotonic() + 60.0
        while len(list(tmp_path.glob("rank*.pid"))) < 3 and time.monotonic() < deadline:
            time.sleep(0.05)
        pids = [int(f.read_text()) for f in sorted(tmp_path.glob("rank*.pid"))]
        assert len(pids) == 3 and parent.poll() is None

        def alive(pid):
            try:
                with open(f"/proc/{pid}/stat") as f:
                    return f.read().rsplit(")", 1)[1].split()[0] != "Z"       # a zombie has exited
            except OSError:
                return False
        assert all(alive(p) for p in pids)
        parent.send_signal(getattr(signal, how))
        rc = parent.wait(timeout=30)
        t_gone = time.monotonic() + 15.0
        while any(alive(p) for p in pids) and time.monotonic() < t_gone:
            time.sleep(0.05)
        assert not any(alive(p) for p in pids), "rank processes survived their parent"
        if how == "SIGKILL":
            assert rc == -signal.SIGKILL
        else:
            assert rc == 128 + int(getattr(signal, how))
            assert parent.stdout.read().strip() == b""
            left = [f for f in set(os.listdir(tempfile.gettempdir())) - before if f.startswith("clive2_bench_id_")]
            assert not left
    finally:
        if parent.poll() is None:
            parent.kill()
        for f in tmp_path.glob("rank*.pid"):
            try:
                os.kill(int(f.read_text()), signal.SIGKILL)
            except (OSError, ValueError):
                pass


def test_bench_single_process_paths_are_unchanged():
    """--gpus 1 does not spawn; under a launcher environment (WORLD_SIZE set) the script is a rank, whatever --gpus says;
    the invalid-render debug bits are refused."""
    import json
    p = _run_bench(["--gpus", "1", "--dry-spawn"])
    out = json.loads(p.stdout)
    assert p.returncode == 0 and out["n_gpus"] == 1 and out["ranks"][0]["explicit_file"] is None
    p = _run_bench(["--gpus", "8", "--dry-spawn"], env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 0 and json.loads(p.stdout)["n_gpus"] == 1
    p = _run_bench(["--debug-flags", "1"])
    assert p.returncode != 0 and "invalid renders" in p.stderr


def test_rendezvous_file_is_private_and_fresh(tmp_path):
    """the id file is created exclusively with mode 0600 and read without following symlinks; a file that
    predates this process (a crashed attempt with the same name) is ignored; the key of the default name changes with
    the launcher's restart count."""
    from clive2_amd import distributed as d
    path = str(tmp_path / "id")
    uid = d.exchange_unique_id(0, 2, lambda: b"k" * 128, 128, path=path)
    assert uid == b"k" * 128 and (os.stat(path).st_mode & 0o777) == 0o600
    assert d.exchange_unique_id(1, 2, None, 128, path=path, timeout=1.0) == uid
    link = str(tmp_path / "link")
    os.symlink(path, link)
    with pytest.raises(TimeoutError):
        d.exchange_unique_id(1, 2, None, 128, path=link, timeout=0.2)       # symlinks are not followed
    old = d._T_IMPORT
    try:
        d._T_IMPORT = os.path.getmtime(path) + d._STALE_SLACK_S + 5.0      # as if this process had started much later
        with pytest.raises(TimeoutError):
            d.exchange_unique_id(1, 2, None, 128, path=path, timeout=0.2)
    finally:
        d._T_IMPORT = old
    env = dict(os.environ)
    try:
        for k in ("CLIVE2_RENDEZVOUS_FILE",):
            os.environ.pop(k, None)
        os.environ["TORCHELASTIC_RESTART_COUNT"] = "0"
        a = d.rendezvous_path()
        os.environ["TORCHELASTIC_RESTART_COUNT"] = "1"
        assert d.rendezvous_path() != a
    finally:
        os.environ.clear(); os.environ.update(env)


def test_two_communicator_bootstraps_over_one_rendezvous_path(tmp_path):
    """bench.py at N > 1 brings up a second communicator (the strong-scaling leg) through the SAME rendezvous file name after
    the first one was destroyed: rank 0 removes the file once `comm_init` has returned on it, and no rank reaches the second
    hand-over before that (the ranks meet rank 0 in collectives in between).  Two hand-overs in a row, three ranks: every rank
    gets the first id, then the second one -- never the first one twice -- and nothing is left behind."""
    import threading
    from clive2_amd import distributed as d
    path = str(tmp_path / "id")
    ids = [b"A" * 128, b"B" * 128]
    got = {r: [] for r in range(3)}
    barrier = threading.Barrier(3)

    def rank(r):
        for k in range(2):
            uid = d.exchange_unique_id(r, 3, lambda: ids[k], 128, path=path, timeout=20.0)
            got[r].append(uid)
            barrier.wait()                   # stands for cl2_comm_init_rank (a collective: returns when every rank has the id)
            d.finish_exchange(r, path)       # rank 0 removes the file
            barrier.wait()                   # the collectives of the workload between the two bootstraps
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert all(got[r] == ids for r in range(3)), got
    assert not os.path.exists(path) and not list(tmp_path.iterdir())
