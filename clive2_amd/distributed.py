"""Multi-GPU sample split: every rank renders its own samples of the replicated
scene; the accumulators -- pure sums (renderer.py:269-273) -- are combined by ONE sum all-reduce
of the packed planar buffer [8][H*W] float32 (66 MB at 1080p).  On GPUs that is an in-place RCCL
all-reduce inside the library (`cl2_comm_init_rank` / `cl2_reduce_accumulators`, include/clive2_amd.h);
this module holds the host side of it: the rank environment, the sample partition, and the hand-over
of RCCL's unique id from rank 0 to the other ranks of the node through a file.  torch is not needed.
"""
import contextlib
import os
import stat
import sys
import tempfile
import time

import numpy as np

_T_IMPORT = time.time()          # close to this rank's process start: files older than that (minus slack) belong to another job
_STALE_SLACK_S = 120.0           # ranks of one job start within this of each other (sequential spawners, slow first imports)


def rendezvous_path():
    """Where rank 0 leaves the communicator id for the other ranks of this job (one node).
    `CLIVE2_RENDEZVOUS_FILE` names it explicitly (any process spawner); otherwise it is derived from what
    all ranks of one launch ATTEMPT share and no other does: MASTER_ADDR/MASTER_PORT, the launcher's run id, its
    restart count (a torchrun restart keeps port, run id and pid) and the launcher's pid (the ranks' common
    parent under `python -m torch.distributed.run`).  `bench.py --gpus N` (self-spawn) passes a fresh random name."""
    explicit = os.environ.get("CLIVE2_RENDEZVOUS_FILE")
    if explicit:
        return explicit
    key = "_".join(str(x) for x in (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"),
                                    os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.getppid(), os.getuid()))
    return os.path.join(os.environ.get("CLIVE2_RENDEZVOUS_DIR", tempfile.gettempdir()), f"clive2_rccl_id_{key}")


def _read_id(path, n_bytes):
    """The id file, or None when it is not (yet) acceptable: it must be a regular file of this user (no symlink is
    followed: the default directory is shared), exactly `n_bytes` long, and not older than this process (minus the
    start-up slack) -- a file left behind by a crashed attempt with the same name predates us."""
    try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_size != n_bytes:
            return None
        if st.st_mtime < _T_IMPORT - _STALE_SLACK_S:
            return None
        uid = os.read(fd, n_bytes + 1)
        return uid if len(uid) == n_bytes else None
    finally:
        os.close(fd)


def exchange_unique_id(rank, world_size, make_id, n_bytes, path=None, timeout=180.0):
    """Rank 0 calls `make_id()` (-> `n_bytes` bytes) and publishes them atomically (exclusive create of a private
    temporary, mode 0600, then rename); the others poll for the file (`_read_id`).  Returns the id on every rank.
    Rank 0 removes the file with `finish_exchange` once every rank has joined the communicator (comm_init returns
    only then)."""
    path = path or rendezvous_path()
    if world_size == 1:
        return make_id()
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != n_bytes:
            raise ValueError(f"unique id has {len(uid)} bytes, expected {n_bytes}")
        tmp = f"{path}.tmp{os.getpid()}"
        try:
            os.unlink(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
        try:
            os.write(fd, uid)
        finally:
            os.close(fd)
        os.replace(tmp, path)              # also replaces whatever a dead attempt left under this name
        return uid
    deadline = time.monotonic() + timeout
    while True:
        uid = _read_id(path, n_bytes)
        if uid is not None:
            return uid
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout:.0f} s (is rank 0 alive?)")
        time.sleep(0.02)


def finish_exchange(rank, path=None):
    if rank == 0:
        try:
            os.unlink(path or rendezvous_path())
        except OSError:
            pass


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on STDOUT when a communicator comes up (seen with 2.27.7: five lines on rank 0).  A rank's
    stdout belongs to the caller -- bench.py prints ONE JSON line there -- so the library's chatter goes to stderr: the file
    descriptor itself is redirected for the duration of the call (the banner is written by C code, not through sys.stdout)."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def join_communicator(renderer, rank, world_size, path=None, timeout=180.0):
    """The whole bootstrap for one rank: id from rank 0, `comm_init` (collective), clean-up."""
    # All ranks of this job are processes of ONE node (the id travels through a local file), so RCCL's own
    # bootstrap sockets can always use the loopback interface -- the container's hostname need not resolve and
    # no other interface need exist.  An explicit NCCL_SOCKET_IFNAME in the environment wins.
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC between the ranks' processes on this pool
    n = renderer._L.cl2_comm_unique_id_bytes()
    with _stdout_to_stderr():
        uid = exchange_unique_id(rank, world_size, renderer.comm_unique_id, n, path=path, timeout=timeout)
    try:
        with _stdout_to_stderr():
            renderer.comm_init(rank, world_size, uid)
    except Exception as e:
        # RCCL's "invalid usage" at this point is almost always two ranks on one device (one GPU per rank is required)
        raise type(e)(f"{e}  [rank {rank} of {world_size} on device {renderer.device}: every rank needs a GPU of its own -- "
                      f"RCCL refuses two ranks on one device]") from e
    finish_exchange(rank, path)


def rank_info():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def samples_for_rank(total_samples, rank, world_size):
    """Split `total_samples` over ranks; the first `total % world` ranks take one extra."""
    base, extra = divmod(int(total_samples), int(world_size))
    return base + (1 if rank < extra else 0)


def radiance_from_packed(packed, height, width):
    """`summed_image / summed_sample_weights` (renderer.py:295-297) from the packed planar buffer."""
    a = np.asarray(packed, dtype=np.float32).reshape(8, height, width)
    img = np.moveaxis(a[0:3], 0, -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nan_to_num(img / a[3][..., None], neginf=0, posinf=0)
