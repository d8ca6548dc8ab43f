"""The collective of SURVEY.md section 8(e) on the ONE GPU a test box has: torch.distributed initialised with backend "nccl" (= RCCL on
ROCm) at world size 1, the first-step contact forces u_opt0 of a device-buffer solve all-gathered through `shard.gather_u0` on the
solve's own HIP stream.  The 2-rank arithmetic of the same function is covered on CPU (tests/test_shard_gloo.py); the 8-GPU curve is the
driver's to measure.  Consumer of the gathered field: g1_mujoco_sim/src/ros_run_simulation.py:214-215."""
import os
import socket

import numpy as np
import pytest

import srbd_oracle as orc

pytestmark = pytest.mark.gpu


def test_rccl_allgather_of_u0_at_world_size_one(built_lib):
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    from g1_locomotion_amd import BatchMPC
    from g1_locomotion_amd.shard import gather_u0, shard_bounds
    import c_oracle
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        B, N = 1024, 10
        lo, hi = shard_bounds(B, dist.get_rank(), dist.get_world_size())
        assert (lo, hi) == (0, B)
        x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=4242, schedule="single")
        d = [torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
        u = torch.zeros((B, N, 12), dtype=torch.float64, device=dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream(device=dev)
        with BatchMPC(horizon=N, max_contacts_per_step=2) as eng:
            for _ in range(3):      # the bench's pattern: solve, then the collective behind it on the same stream, nothing waited for in between
                eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), status=st.data_ptr(),
                                 stream=stream.cuda_stream)
                with torch.cuda.stream(stream):
                    allu = gather_u0(u[:, 0, :])
            stream.synchronize()
            torch.cuda.synchronize(dev)
        assert allu.shape == (B, 12) and allu.is_cuda
        assert torch.equal(allu, u[:, 0, :])
        # and they are the forces of the QPs: against the compiled oracle
        ref = c_oracle.solve_batch(orc.default_params(N), x0, xr, ft, ct)
        ok = (st.cpu().numpy() == orc.STATUS_SOLVED) & (ref["status"] == orc.STATUS_SOLVED)
        assert ok.mean() > 0.98
        assert np.abs(allu.cpu().numpy()[ok] - ref["u"][ok, 0, :]).max() < 2e-3
    finally:
        dist.destroy_process_group()

