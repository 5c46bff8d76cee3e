"""world_size-2 CPU (gloo) test of the multi-rank path: slabs are sharded across ranks with no
collective on the data path, so every rank's result must equal its slice of the unsharded
result bit for bit; the timing reduction takes the slowest rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import TINY_LEVELS, random_case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here, os.path.join(here, "golden")]
    from opus_amd import dist as odist
    from oracle import msmv_oracle

    dist = odist.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    feats, loc, w, grad_out = random_case(42, TINY_LEVELS, B=5, N=6, C=8, Q=7, P=2)
    full_out = msmv_oracle.forward(feats, loc, w)
    full_df, full_dloc, full_dw = msmv_oracle.backward(grad_out, feats, loc, w)

    lo, hi = odist.shard_range(5, world, rank)
    sh = odist.shard_slabs([torch.from_numpy(x) for x in feats + [loc, w, grad_out]], world, rank)
    sh = [t.numpy() for t in sh]
    f_sh, loc_sh, w_sh, g_sh = sh[:4], sh[4], sh[5], sh[6]
    out = msmv_oracle.forward(f_sh, loc_sh, w_sh)
    df, dloc, dw = msmv_oracle.backward(g_sh, f_sh, loc_sh, w_sh)
    ok = (np.array_equal(out, full_out[lo:hi]) and np.array_equal(dloc, full_dloc[lo:hi])
          and np.array_equal(dw, full_dw[lo:hi])
          and all(np.array_equal(a, b[lo:hi]) for a, b in zip(df, full_df)))

    odist.fence(dist)
    slowest = odist.max_over_ranks(1.0 + rank, dist, torch.device("cpu"))
    flags = torch.tensor([int(ok), hi - lo], dtype=torch.int64)
    gathered = [torch.zeros_like(flags) for _ in range(world)]
    dist.all_gather(gathered, flags)
    if rank == 0:
        with open(os.path.join(tmpdir, "result.txt"), "w") as f:
            f.write(f"{slowest} {' '.join(str(int(g[0])) for g in gathered)} "
                    f"{sum(int(g[1]) for g in gathered)}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_is_exact(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    slowest, ok0, ok1, nslabs = open(tmp_path / "result.txt").read().split()
    assert float(slowest) == 2.0          # max over ranks of (1 + rank)
    assert ok0 == "1" and ok1 == "1"
    assert int(nslabs) == 5               # the shards cover every slab exactly once


def test_shard_range_partitions():
    from opus_amd.dist import shard_range
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


# ---- bench.py's own multi-rank control flow, compute stubbed (no GPU): launch -> per-rank seeds
# and slab ranges -> fence -> slowest rank's time -> ONE line from rank 0
def _bench_line(cmd):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable] + cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_stubbed(launcher):
    args = ["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--stub-compute", "--workload", "L-fwdbwd"]
    if launcher == "torchrun":
        args = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] + args
    r = _bench_line(args)
    assert r["n_gpus"] == 2 and r["steps"] == 6 and r["warmup"] == 2 and r["data"] == "stub"
    assert r["scaling"] == "weak" and r["config"]["backend"] == "gloo"
    assert [(x["rank"], x["seed"], x["slabs"]) for x in r["config"]["ranks"]] == \
        [(0, 1234, [0, 32]), (1, 1235, [32, 64])]
    # rank 1 sleeps 4 ms per step, rank 0 only 2 ms: the job's time is the slowest rank's
    assert r["ms_per_step"] >= 4.0
    # value = points of ALL ranks / that time
    assert abs(r["value"] - 2 * 32 * 4800 * 2 / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]


def test_bench_split_sample_and_allreduce_stubbed():
    r = _bench_line(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-compute",
                     "--workload", "T-fwd", "--split-sample"])
    assert r["scaling"] == "strong"
    assert [x["slabs"] for x in r["config"]["ranks"]] == [[0, 16], [16, 32]]
    assert abs(r["value"] - 32 * 600 * 4 / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    # the training configuration's gradient all-reduce (12 buckets of 25 MB) between two ranks
    r = _bench_line(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-compute",
                     "--workload", "train-M"])
    assert r["config"]["points_per_step_per_gpu"] == 6 * 32 * 2400 * 2 and r["n_gpus"] == 2
    # two buckets are handed to the collective after each of the six layers' backward, in layer order
    # (not all twelve behind the whole step), and the cost of the collective is reported
    assert r["config"]["allreduce_schedule"] == [[layer, 2 * layer + k] for layer in range(6) for k in range(2)]
    assert r["config"]["allreduce_bytes_per_step"] == 300 * 1000 * 1000
    assert r["allreduce_exposed_us"] is not None and r["ms_per_step_without_allreduce"] >= 4.0


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--stub-compute"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "does not match" in p.stderr


def test_spawn_ranks_reports_a_failing_rank():
    import sys
    from opus_amd.dist import spawn_ranks
    code = "import os, sys, time; r = int(os.environ['RANK']); time.sleep(0.2 if r else 30); sys.exit(7 if r else 0)"
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    try:
        assert spawn_ranks(2, [f.name]) == 7          # rank 1 fails; rank 0 (asleep) is terminated, not awaited
    finally:
        os.unlink(f.name)


def _rank_script(code):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    return f.name


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie still answers signal 0: look at its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_spawn_ranks_ends_a_rank_that_never_returns(tmp_path):
    """A rank stuck for ever (here: asleep, and deaf to SIGTERM like a process blocked in a driver
    call) must not hold the launcher: after the deadline it is terminated, then killed, and the
    launcher returns 124 with no child left behind."""
    import time
    from opus_amd.dist import spawn_ranks
    pidfile = tmp_path / "pids"
    code = ("import os, signal, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            f"open({str(pidfile)!r}, 'a').write(str(os.getpid()) + '\\n')\n"
            "if r == 1:\n"
            "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
            "    time.sleep(600)\n"
            "sys.exit(0)\n")
    name = _rank_script(code)
    try:
        t0 = time.monotonic()
        assert spawn_ranks(2, [name], timeout_s=3.0, grace_s=1.0) == 124
        assert time.monotonic() - t0 < 30
        pids = [int(x) for x in pidfile.read_text().split()]
        assert len(pids) == 2 and not any(_alive(p) for p in pids)
    finally:
        os.unlink(name)


def test_spawn_ranks_leaves_no_child_behind_when_interrupted(tmp_path):
    """An exception in the launcher (KeyboardInterrupt while it polls) ends the ranks it started."""
    import subprocess
    import sys
    import time
    pidfile = tmp_path / "pids"
    rank = _rank_script("import os, time\n"
                        f"open({str(pidfile)!r}, 'a').write(str(os.getpid()) + '\\n')\n"
                        "time.sleep(600)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    parent = ("import sys\n"
              f"sys.path.insert(0, {root!r})\n"
              "from opus_amd.dist import spawn_ranks\n"
              f"sys.exit(spawn_ranks(2, [{rank!r}], timeout_s=600, grace_s=1.0))\n")
    try:
        p = subprocess.Popen([sys.executable, "-c", parent])
        for _ in range(300):
            if pidfile.exists() and len(pidfile.read_text().split()) == 2:
                break
            time.sleep(0.1)
        import signal
        p.send_signal(signal.SIGINT)
        p.wait(timeout=60)
        pids = [int(x) for x in pidfile.read_text().split()]
        assert len(pids) == 2 and not any(_alive(q) for q in pids)
    finally:
        os.unlink(rank)
