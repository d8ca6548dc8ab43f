"""N > 1 path on CPU: two gloo ranks shard a batch, solve their halves with the CPU oracle (standing in for the
per-rank GPU solve), all-gather u_opt0 exactly as bench.py does over RCCL, and must reproduce the single-process
result.  Also the shard arithmetic."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_the_batch():
    from g1_locomotion_amd.shard import shard_bounds
    for total in (0, 1, 7, 4096, 524288):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import srbd_oracle as orc
    import c_oracle
    from g1_locomotion_amd.shard import shard_bounds, gather_u0
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 16
    x0, xr, ft, ct = orc.synthetic_batch(B, 10, 1234, "single")
    lo, hi = shard_bounds(B, rank, world)
    out = c_oracle.solve_batch(orc.SrbdParams(), x0[lo:hi], xr[lo:hi], ft[lo:hi], ct[lo:hi])
    u0 = torch.from_numpy(np.ascontiguousarray(out["u"][:, 0, :]))
    allu = gather_u0(u0)
    dist.barrier()
    if rank == 0:
        q.put(allu.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_allgather_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import srbd_oracle as orc
    import c_oracle
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x0, xr, ft, ct = orc.synthetic_batch(16, 10, 1234, "single")
    ref = c_oracle.solve_batch(orc.SrbdParams(), x0, xr, ft, ct)["u"][:, 0, :]
    np.testing.assert_array_equal(got, ref)
