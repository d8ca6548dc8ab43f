"""HBM roofline of the two element-wise cascade kernels (include/srbdqp_cascade.h) on device-resident buffers.

    python tools/cascade_bench.py [items]

Algorithmic bytes per item: swing 104 B (64 in, 40 out; coefficients not stored), wbid_reference 488 B (296 in, 192 out).
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

    out = {"items": B}
    for name, fn, bytes_per in (("swing_f64", run_swing, 104), ("wbid_reference_f64", run_wbid, 488)):
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
        gbps = B * bytes_per / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "items_per_s": B / (ms * 1e-3), "bytes_per_item": bytes_per, "achieved_GBps": gbps, "peak_GBps": 8000.0, "frac": gbps / 8000.0}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
