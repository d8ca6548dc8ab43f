"""profiles/<round>_<kernel>_pmc_summary.json from tools/pmc_collect.sh output + the rocprofv3 --stats csv.
    python tools/pmc_summary.py gpurun_out/pmc_dir gpurun_out/stats_dir profiles/r01_compact_n10_s2_pmc_summary.json"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import subprocess


def main():
    pmc_dir, stats_dir, out = sys.argv[1:4]
    c = json.loads(subprocess.check_output([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_parse.py"), pmc_dir]))
    avg_us = calls = None
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "srbdqp_compact_kernel" in row["Name"]:
                avg_us, calls = float(row["AverageNs"]) / 1e3, int(row["Calls"])
    B, N = 4096, 10
    alg = B * ((13 + 13 * N + 12 * N + 4 * N) * 8 + (12 * N + 13 * (N + 1)) * 8 + 8)
    fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    simd_quads = 1024 * xcd_cycles / 4.0
    d = {"round": 1, "kernel": "srbdqp_compact_kernel<10,2>", "bench_kernel_name": "compact_f64_n10_s2", "batch_per_launch": B,
         "workload": "bench.py --streams 1 (configs[1]: B=4096, N=10, 2-contact, fp64; longest-first hint on); one launch at a time",
         "command": "tools/pmc_collect.sh (one rocprofv3 --pmc <group> --kernel-trace pass per counter group) + rocprofv3 --kernel-trace --stats on the same bench command",
         "kernel_avg_us_rocprof_kernel_trace": avg_us, "kernel_calls": calls,
         "counters_mean_per_dispatch": {k: v for k, v in c.items() if not k.startswith("_")},
         "hbm": {"FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
                 "fetch_bytes_corrected_x2": None if fetch is None else fetch * 1024 * 2, "write_bytes": None if write is None else write * 1024,
                 "traffic_bytes_per_launch": None if fetch is None or write is None else fetch * 2048 + write * 1024,
                 "algorithmic_bytes_per_launch": alg,
                 "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE taken as is"},
         "clock_GHz": None if avg_us is None else xcd_cycles / (avg_us * 1e3),
         "per_qp": {"valu_insts": c["SQ_INSTS_VALU"] / B, "salu_insts": c["SQ_INSTS_SALU"] / B, "lds_insts": c["SQ_INSTS_LDS"] / B, "mfma_insts": c["SQ_INSTS_MFMA"] / B,
                    "fma_f64": c["SQ_INSTS_VALU_FMA_F64"] / B, "add_f64": c["SQ_INSTS_VALU_ADD_F64"] / B, "mul_f64": c["SQ_INSTS_VALU_MUL_F64"] / B},
         "valu_util": c["SQ_ACTIVE_INST_VALU"] / simd_quads,
         "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * xcd_cycles),
         "lds_busy": c["SQ_LDS_IDX_ACTIVE"] / (256 * xcd_cycles), "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
         "wave_time_split": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                             "parked_waitcnt_or_barrier": 1.0 - (c["SQ_ACTIVE_INST_ANY"] + c["SQ_WAIT_INST_ANY"]) / c["SQ_WAVE_CYCLES"]},
         "notes": "SQ_* are summed over the chip; SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* are in 4-cycle quads; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
                  "valu_util = VALU-issuing quads / (1024 SIMDs x kernel quads); mfma_util = matrix-pipe busy cycles / SIMD cycles."}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps({k: d[k] for k in ("kernel_avg_us_rocprof_kernel_trace", "clock_GHz", "valu_util", "mfma_util", "lds_busy", "wave_time_split", "per_qp")}, indent=1))
    print(d["hbm"])


if __name__ == "__main__":
    main()
