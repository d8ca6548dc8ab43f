#!/usr/bin/env python3
"""Where a SHORT timed region of bench.py goes (the driver times 20 steps after 5 warm-up steps; the default run 200).

    python tools/region_overhead.py [reps]

Times regions of K = 5, 10, 20, 40, 80, 200 steps of configs[1] exactly the way bench.py does (sync, K steps on 2 streams with the
longest-first hint, sync) and fits elapsed(K) = a + b K: b is the steady step time, a the fixed cost of a region (launch ramp, drain
of the last launch, the wake-up of the synchronising host thread).  Also the host's own time to ISSUE the K steps (no sync)."""
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    import torch
    args = bench.parse_args([])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    leg = bench.Leg(1, bench.CONFIGS[1]["batch"], args, 0, 0, dev, torch, bench.NBATCH, max_streams=2)
    for i in range(16):
        leg.step(i, S=2, hint="none" if i < leg.NO else "own")
    torch.cuda.synchronize(dev)
    base = 16
    Ks = [5, 10, 20, 40, 80, 200]
    res = {}
    for K in Ks:
        el, iss = [], []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(base, base + K):
                leg.step(k, S=2, hint="own")
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            el.append(t2 - t0); iss.append(t1 - t0)
        res[K] = (float(np.median(el)), float(np.min(el)), float(np.median(iss)))
        B = leg.B
        print("K=%4d  elapsed median %8.1f us (min %8.1f)  = %6.2f us/step = %5.2f M QP/s   host issue %7.1f us (%.1f us/step)"
              % (K, res[K][0] * 1e6, res[K][1] * 1e6, res[K][0] * 1e6 / K, B * K / res[K][0] / 1e6, res[K][2] * 1e6, res[K][2] * 1e6 / K))
    x = np.array(Ks, float); y = np.array([res[K][0] for K in Ks]) * 1e6
    b, a = np.polyfit(x, y, 1)
    print("fit: elapsed(K) = %.1f us + %.2f us x K   (steady %.2f M QP/s; a region of 20 steps loses %.1f %% to the fixed part)"
          % (a, b, leg.B / b, 100 * a / (a + 20 * b)))
    leg.close()


if __name__ == "__main__":
    main()
