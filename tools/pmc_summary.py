"""profiles/<round>_<kernel>_pmc_summary.json from tools/pmc_collect.sh output + the rocprofv3 --stats csv.
    python tools/pmc_summary.py gpurun_out/pmc_dir gpurun_out/stats_dir profiles/r01_compact_n10_s2_pmc_summary.json"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import subprocess


def main():
    pmc_dir, stats_dir, out = sys.argv[1:4]
    from pmc_parse import parse
    per = parse(pmc_dir, ["srbdqp_setup1_kernel", "srbdqp_compact_kernel", "srbdqp_admm_kernel"])
    stats = {}
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            for key in per:
                if key in row["Name"]:
                    stats[key] = {"name": row["Name"], "avg_us": float(row["AverageNs"]) / 1e3, "calls": int(row["Calls"])}
    B, N = 4096, 10
    alg = B * ((13 + 13 * N + 12 * N + 4 * N) * 8 + (12 * N + 13 * (N + 1)) * 8 + 8)
    kernels = {}
    tot_fetch = tot_write = tot_us = 0.0
    for key, c in per.items():
        xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        simd_quads = 1024 * xcd_cycles / 4.0
        fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        k = {"rocprof_kernel_trace": stats.get(key),
             "counters_mean_per_dispatch": {n: v for n, v in c.items() if not n.startswith("_")},
             "hbm_bytes_per_launch": {"fetch_x2": fetch * 2048, "write": write * 1024},
             "per_qp": {"valu_insts": c["SQ_INSTS_VALU"] / B, "salu_insts": c["SQ_INSTS_SALU"] / B, "lds_insts": c["SQ_INSTS_LDS"] / B, "mfma_insts": c["SQ_INSTS_MFMA"] / B,
                        "fma_f64": c["SQ_INSTS_VALU_FMA_F64"] / B, "add_f64": c["SQ_INSTS_VALU_ADD_F64"] / B, "mul_f64": c["SQ_INSTS_VALU_MUL_F64"] / B},
             "valu_util": c["SQ_ACTIVE_INST_VALU"] / simd_quads, "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * xcd_cycles),
             "lds_busy": c["SQ_LDS_IDX_ACTIVE"] / (256 * xcd_cycles), "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
             "wave_time_split": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                 "parked_waitcnt_or_barrier": 1.0 - (c["SQ_ACTIVE_INST_ANY"] + c["SQ_WAIT_INST_ANY"]) / c["SQ_WAVE_CYCLES"]}}
        if key in stats:
            k["clock_GHz"] = xcd_cycles / (stats[key]["avg_us"] * 1e3)
            tot_us += stats[key]["avg_us"]
        tot_fetch += fetch * 2048; tot_write += write * 1024
        kernels[key] = k
    name = ("split_f64_n10_s2" if len(per) > 1 else "wave_f64_n10_s2" if "srbdqp_setup1_kernel" in per else "compact_f64_n10_s2")
    d = {"round": 1, "bench_kernel_name": name,
         "pipeline": {"split_f64_n10_s2": "set-up kernel + srbdqp_admm_kernel<10,2> (ADMM + roll-out), K^-1 handed over through HBM",
                      "wave_f64_n10_s2": "srbdqp_setup1_kernel<10,2,true>: the whole solve on one wave per QP",
                      "compact_f64_n10_s2": "srbdqp_compact_kernel<10,2>: the whole solve on 4 waves per QP"}[name],
         "batch_per_launch": B,
         "workload": "bench.py --streams 1 (configs[1]: B=4096, N=10, 2-contact, fp64; longest-first hint on); one launch at a time",
         "command": "tools/pmc_collect.sh (one rocprofv3 --pmc <group> --kernel-trace pass per counter group) + rocprofv3 --kernel-trace --stats on the same bench command",
         "kernel_avg_us_rocprof_kernel_trace": tot_us,
         "hbm": {"fetch_bytes_corrected_x2": tot_fetch, "write_bytes": tot_write, "traffic_bytes_per_launch": tot_fetch + tot_write,
                 "algorithmic_bytes_per_launch": alg,
                 "handover_bytes_per_launch_expected": None,
                 "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE taken as is. "
                         "In the split pipeline the traffic above the algorithmic bytes is the hand-over of K^-1 and the persistent strip between the two kernels (written once, read once); the wave and compact kernels have no hand-over."},
         "kernels": kernels,
         "notes": "SQ_* are summed over the chip; SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* are in 4-cycle quads; GRBM_GUI_ACTIVE is summed over the 8 XCDs."}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps({"avg_us_sum": tot_us, "hbm": d["hbm"], "kernels": {k: {q: v[q] for q in ("valu_util", "mfma_util", "lds_busy", "wave_time_split", "per_qp", "rocprof_kernel_trace")} for k, v in kernels.items()}}, indent=1))


if __name__ == "__main__":
    main()
