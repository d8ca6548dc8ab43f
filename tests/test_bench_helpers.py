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


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it: the script starts one process per rank before anything
    touches a GPU, rank 0 prints the one JSON line, both the with- and the without-all-gather rates are reported.  Here on
    CPU: gloo, the solve stubbed (the launcher path, the rotation over batches and the collective are what is tested)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-solve", "--backend", "gloo", "--steps", "6",
                        "--warmup", "2", "--batch", "32"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["unit"] == "QP/s"
    assert d["config"]["allgather_in_value"] is True and d["config"]["distinct_batches"] == 4
    assert d["value"] > 0 and d["value_without_allgather"] > 0
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-solve", "--backend", "gloo", "--steps", "3",
                         "--warmup", "1", "--batch", "16", "--no-allgather"], capture_output=True, text=True, timeout=300, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["config"]["allgather_in_value"] is False and d2["value_with_allgather"] > 0
