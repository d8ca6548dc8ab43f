"""HBM roofline of the element-wise cascade kernels (include/srbdqp_cascade.h) on device-resident buffers.

    python tools/cascade_bench.py [items]

Algorithmic bytes per item: swing 104 B (64 in, 40 out; coefficients not stored), wbid_reference 488 B (296 in, 192 out),
mpc_inputs 2541 B per robot at N = 10 (225 in; 10 x 228 + 24 out; run on items / 16 robots).
Prints one JSON object with the achieved GB/s and the fraction of the 8 TB/s HBM3E peak.
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024
    import torch
    from g1_locomotion_amd import BatchMPC
    eng = BatchMPC(horizon=10)
    lib, h = eng._lib, eng._h
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    r = lambda *s: torch.rand(*s, device=dev, dtype=torch.float64, generator=g)
    ps, pf, zm, t = r(B, 3), r(B, 3), r(B), r(B)
    pos, vz, az = torch.empty(B, 3, device=dev, dtype=torch.float64), torch.empty(B, device=dev, dtype=torch.float64), torch.empty(B, device=dev, dtype=torch.float64)
    x, u, ft = r(B, 13), r(B, 12), r(B, 12)
    R, bv, ba, ca = (torch.empty(B, k, device=dev, dtype=torch.float64) for k in (9, 6, 6, 3))
    st = torch.cuda.Stream()
    p = lambda a: C.c_void_p(a.data_ptr())
    sp = C.c_void_p(st.cuda_stream)

    def run_swing():
        rc = lib.srbdqp_swing_device_f64(h, B, p(ps), p(pf), p(zm), p(t), -0.02, 0.8, p(pos), p(vz), p(az), None, sp)
        assert rc == 0

    def run_wbid():
        rc = lib.srbdqp_wbid_reference_device_f64(h, B, p(x), p(u), p(ft), 1, p(R), p(bv), p(ba), p(ca), sp)
        assert rc == 0

    from g1_locomotion_amd import _lib
    Bi = max(B // 16, 1)                                     # robots for the step before the QP (N = 10 horizon rows each)
    N = eng.N
    x0i, fti, sti, vri = r(Bi, 13), r(Bi, 12), r(Bi) * 5.0, r(Bi, 2) - 0.5
    xri, fi, pci = (torch.empty(Bi, N, k, device=dev, dtype=torch.float64) for k in (13, 12, 3))
    lpi = torch.empty(Bi, 3, device=dev, dtype=torch.float64)
    cti = torch.empty(Bi, N, 4, device=dev, dtype=torch.uint8)
    gait = _lib.Gait()
    gait.struct_size, gait.period_steps, gait.double_support_steps = C.sizeof(_lib.Gait), 6, 1
    gait.com_target[0], gait.com_target[1], gait.com_target[2], gait.hip_offset_y = 0.05268, 7.44e-5, 0.59798, 0.0645

    def run_inputs():
        rc = lib.srbdqp_mpc_inputs_device_f64(h, Bi, p(x0i), p(fti), p(sti), p(vri), None, C.byref(gait), p(xri), p(fi), p(cti), p(pci), p(lpi), sp)
        assert rc == 0

    out = {"items": B}
    for name, fn, bytes_per, nitems in (("swing_f64", run_swing, 104, B), ("wbid_reference_f64", run_wbid, 488, B),
                                        ("mpc_inputs_f64", run_inputs, 225 + N * 228 + 24, Bi)):
        with torch.cuda.stream(st):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            reps = 10
            for _ in range(reps):
                fn()
            e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbps = nitems * bytes_per / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "items": nitems, "items_per_s": nitems / (ms * 1e-3), "bytes_per_item": bytes_per, "achieved_GBps": gbps, "peak_GBps": 8000.0, "frac": gbps / 8000.0}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
