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



def test_c_abi_gather_over_an_rccl_communicator_of_the_callers(built_lib):
    """srbdqp_gather_u0_f64 (include/srbdqp.h): the all-gather of u_opt0 for consumers without torch.distributed -- the caller hands over an ncclComm_t it made
    itself.  Here: a communicator of one rank made through RCCL's own C API (ncclGetUniqueId / ncclCommInitRank, bound with ctypes), a solve on a stream, the gather
    on the same stream; the gathered field equals the first-step forces of the solve and the C oracle's.  (srbdqp_shard_range is covered on the CPU.)"""
    import ctypes as C
    import torch
    import c_oracle
    from g1_locomotion_amd import BatchMPC, _lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError as e:
        pytest.skip("no librccl.so.1 to make a communicator with (%s)" % e)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        N, B = 10, 2048
        x0, xr, ft, ct = orc.synthetic_batch(B, N, seed=77, schedule="single")
        d = [torch.from_numpy(v).to(dev) for v in (x0, xr, ft, ct)]
        u = torch.zeros((B, N, 12), dtype=torch.float64, device=dev)
        u0_all = torch.full((B, 12), -1.0, dtype=torch.float64, device=dev)
        st = torch.cuda.Stream(device=dev)
        with BatchMPC(horizon=N, max_contacts_per_step=2) as eng:
            eng.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), u.data_ptr(), stream=st.cuda_stream)
            rc = built_lib.srbdqp_gather_u0_f64(eng._h, C.c_void_p(u.data_ptr()), B, C.c_void_p(u0_all.data_ptr()), comm, C.c_void_p(st.cuda_stream))
            assert rc == 0, built_lib.srbdqp_last_error(eng._h)
            st.synchronize()
            assert torch.equal(u0_all, u[:, 0, :])
            # a null communicator is refused, not dereferenced
            assert built_lib.srbdqp_gather_u0_f64(eng._h, C.c_void_p(u.data_ptr()), B, C.c_void_p(u0_all.data_ptr()), None, None) == _lib.E_INVALID
        ref = c_oracle.solve_batch(orc.default_params(N), x0[:64], xr[:64], ft[:64], ct[:64])
        ok = ref["status"] == orc.STATUS_SOLVED
        assert ok.mean() > 0.9 and np.abs(u0_all[:64].cpu().numpy()[ok] - ref["u"][ok, 0, :]).max() < 2e-3
    finally:
        rccl.ncclCommDestroy(comm)
