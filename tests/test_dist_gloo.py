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
