"""One-process-per-GPU helpers for the sampling op.

The op is embarrassingly parallel over its batch dimension B' = B*T*G: every slab reads only
its own feature maps, locations and weights and writes only its own outputs / gradients
(reference kernels: msmv_sampling_forward.cu:96-101,127; msmv_sampling_backward.cu:135-142).
Multi-GPU execution therefore shards slabs (or whole samples, as the reference's DDP does,
train.py:93) across ranks with NO collective on the data path; torch.distributed (RCCL on
ROCm) is used only for rendezvous, barriers, timing reductions and - in a training
configuration - the parameter-gradient all-reduce that lives outside the op.
"""
import os
import socket
import subprocess
import sys
import time

import torch


def env_rank():
    """(rank, world_size, local_rank) from the launcher's environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None, force=False):
    """Initialise the default process group if WORLD_SIZE > 1; returns torch.distributed or None.
    ``force``: also with one rank (a group of one: every collective then goes through the backend's
    library and returns its input - a rehearsal of the calls, e.g. of RCCL on a one-GPU machine)."""
    rank, world, _ = env_rank()
    if world == 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", "1")
    if not dist.is_initialized():
        kwargs = {}
        if device is not None and backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend, **kwargs)
    return dist


def shard_range(n_items, world, rank):
    """Contiguous, balanced [start, stop) of `n_items` for `rank`: the first n_items % world
    ranks get one extra item; ranks beyond n_items get an empty range."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_slabs(tensors, world, rank):
    """Slice every tensor's leading (slab) dimension to this rank's range."""
    n = tensors[0].shape[0]
    lo, hi = shard_range(n, world, rank)
    return [t[lo:hi].contiguous() for t in tensors]


def max_over_ranks(value, dist, device):
    """The slowest rank's value (e.g. elapsed seconds)."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def fence(dist, device=None):
    """synchronize + barrier + synchronize: both sides of a timed region."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(world, argv, env=None, poll_s=0.2, timeout_s=None, grace_s=5.0):
    """Start `world` fresh interpreter processes running `argv` (script + arguments), one per
    rank, with the environment torch.distributed.run would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port), and wait for all of them.

    For a caller that was started WITHOUT a launcher although it is to use several GPUs.  The
    caller must not have touched the GPU: the children are new processes (fork + exec of the
    interpreter), never a re-exec of this one.  Their stdout / stderr are the caller's, so the
    one line rank 0 prints is this process's output.  Returns the exit status to leave with:
    0 if every rank returned 0, else the first failing rank's status; when one rank fails, the
    others (exactly the processes started here) are ended instead of left waiting in a collective.

    `timeout_s` (default: OPUS_RANK_TIMEOUT_S or 600) is the deadline for the whole job: a rank that
    is stuck - in a collective whose peer never arrives, in a driver call - would otherwise hold
    this process for ever.  On expiry every rank still running is ended and the status is 124.
    Ending a rank is SIGTERM, then SIGKILL after `grace_s` seconds (a process blocked inside a
    collective or the GPU driver may not act on the first).  Whatever ends this function - return,
    exception, KeyboardInterrupt - no child started here is left running."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("OPUS_RANK_TIMEOUT_S", "600"))
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    procs = []

    def end_all():
        running = [p for p in procs if p.poll() is None]
        for p in running:
            p.terminate()
        t_kill = time.monotonic() + grace_s
        for p in running:
            try:
                p.wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        for p in running:
            p.wait()

    status = 0
    try:
        for rank in range(world):
            e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
        deadline = time.monotonic() + timeout_s
        live = set(range(world))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
            if status != 0 and live:
                end_all()                    # the survivors would wait for the failed rank for ever
                live.clear()
            elif live and time.monotonic() > deadline:
                print(f"spawn_ranks: rank(s) {sorted(live)} still running after {timeout_s:g} s - ending them",
                      file=sys.stderr, flush=True)
                end_all()
                live.clear()
                status = 124
            elif live:
                time.sleep(poll_s)
    finally:
        end_all()
    return status
