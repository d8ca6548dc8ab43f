"""bench.py bookkeeping: the algorithmic flop / byte model is SURVEY.md section 8(d)'s."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_flops_and_bytes_match_the_survey():
    b = _bench()
    assert abs(b.algorithmic_flops(10, 50) / 1e6 - 4.29) < 0.01          # N=10, K=50 -> 4.29 MF
    assert abs(b.algorithmic_flops(20, 50) / 1e6 - 26.6) < 0.1
    for N, mf in ((8, 2.46), (12, 6.85), (16, 14.6), (24, 43.9)):
        assert abs(b.algorithmic_flops(N, 50) / 1e6 - mf) < 0.06 * mf
    assert b.algorithmic_bytes(10) == 4536                               # 4,536 B per QP (N=10, fp64)
    assert b.PEAK_FP64_TFLOPS == 78.6 and b.BATCH_PER_GPU == 4096 and b.HORIZON == 10
