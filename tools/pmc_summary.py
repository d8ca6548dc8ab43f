"""profiles/<round>_<kernel>_pmc_summary.json from tools/pmc_collect.sh output + the rocprofv3 --stats csv of the same bench
command.  Kernel-agnostic: the dominant solver kernel is the srbdqp:: kernel with the largest total time in the stats.
    python tools/pmc_summary.py gpurun_out/pmc_dir gpurun_out/stats_dir profiles/r02_<name>_pmc_summary.json <bench kernel name> <B> <N> <esz> [round]

`executed`: the flops the kernel really issues per launch = (FMA x 2 + ADD + MUL) x 64 lanes (fp64 and fp32 counters, wave
instructions, EXEC-masked lanes included) + MFMA instructions x 2048 (16x16x4 fp64 / fp32); `bound`: what the counters say."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PEAK = {8: 78.6, 4: 157.3}


def main():
    pmc_dir, stats_dir, out, kname, B, N, esz = sys.argv[1:8]
    B, N, esz = int(B), int(N), int(esz)
    rnd = int(sys.argv[8]) if len(sys.argv) > 8 else 2
    lps = float(sys.argv[9]) if len(sys.argv) > 9 else 1.0     # launches of the dominant kernel per solve (rho restart on: 2)
    from pmc_parse import parse
    # kernels in the stats: pick the srbdqp solver kernels (not the 1-workgroup helper kernels)
    stats = {}
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "srbdqp::srbdqp_" in row["Name"] and "schedule_kernel" not in row["Name"] and "restart_select" not in row["Name"]:
                stats[row["Name"]] = {"name": row["Name"], "avg_us": float(row["AverageNs"]) / 1e3, "calls": int(row["Calls"]),
                                      "total_us": float(row["TotalDurationNs"]) / 1e3 if "TotalDurationNs" in row else float(row["AverageNs"]) / 1e3 * int(row["Calls"])}
    if not stats:
        raise SystemExit("no srbdqp kernel in the stats csv")
    keys = sorted(stats, key=lambda k: -stats[k]["total_us"])
    short = {k: k.split("srbdqp::")[1].split("(")[0] for k in keys}
    per = parse(pmc_dir, [short[k] for k in keys])
    alg = B * ((13 + 13 * N + 12 * N + 4 * N) * esz + (12 * N + 13 * (N + 1)) * esz + 8)
    kernels = {}
    tot_fetch = tot_write = tot_us = tot_flops = 0.0
    dom = None
    for k in keys:
        c = per.get(short[k])
        if not c:
            continue
        g = lambda n: c.get(n, 0.0)
        xcd_cycles = g("GRBM_GUI_ACTIVE") / 8.0
        simd_quads = 1024 * xcd_cycles / 4.0
        fl64 = (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")) * 64
        fl32 = (2 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32")) * 64
        # SQ_INSTS_VALU_FLOPS_FP32/64 (flops per lane, summed over wave instructions) count a packed v_pk_fma_f32 as 4: the
        # instruction counters above see it once.  Checked on the fp64 one-wave kernel (no packed code): both agree to 1 %.
        if g("SQ_INSTS_VALU_FLOPS_FP32") > 0: fl32 = g("SQ_INSTS_VALU_FLOPS_FP32") * 64
        if g("SQ_INSTS_VALU_FLOPS_FP64") > 0: fl64 = g("SQ_INSTS_VALU_FLOPS_FP64") * 64
        flmf = g("SQ_INSTS_MFMA") * 2048
        kk = {"rocprof_kernel_trace": stats[k],
              "counters_mean_per_dispatch": {n: v for n, v in c.items() if not n.startswith("_")},
              "hbm_bytes_per_launch": {"fetch_x2": g("FETCH_SIZE") * 2048, "write": g("WRITE_SIZE") * 1024},
              "per_qp": {"valu_insts": g("SQ_INSTS_VALU") / B, "salu_insts": g("SQ_INSTS_SALU") / B, "lds_insts": g("SQ_INSTS_LDS") / B,
                         "mfma_insts": g("SQ_INSTS_MFMA") / B, "fma_f64": g("SQ_INSTS_VALU_FMA_F64") / B, "add_f64": g("SQ_INSTS_VALU_ADD_F64") / B,
                         "mul_f64": g("SQ_INSTS_VALU_MUL_F64") / B, "fma_f32": g("SQ_INSTS_VALU_FMA_F32") / B, "add_f32": g("SQ_INSTS_VALU_ADD_F32") / B,
                         "mul_f32": g("SQ_INSTS_VALU_MUL_F32") / B, "int32": g("SQ_INSTS_VALU_INT32") / B, "cvt": g("SQ_INSTS_VALU_CVT") / B},
              "executed_flops_per_launch": {"valu_f64": fl64, "valu_f32": fl32, "mfma": flmf, "total": fl64 + fl32 + flmf},
              "valu_util": g("SQ_ACTIVE_INST_VALU") / simd_quads if simd_quads else None,
              "mfma_util": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * xcd_cycles) if xcd_cycles else None,
              "lds_busy": g("SQ_LDS_IDX_ACTIVE") / (256 * xcd_cycles) if xcd_cycles else None,
              "lds_bank_conflict_share": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
              "wave_time_split": {"issuing": g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "issue_stalled": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0),
                                  "parked_waitcnt_or_barrier": 1.0 - (g("SQ_ACTIVE_INST_ANY") + g("SQ_WAIT_INST_ANY")) / max(g("SQ_WAVE_CYCLES"), 1.0)}}
        kk["clock_GHz"] = xcd_cycles / (stats[k]["avg_us"] * 1e3) if xcd_cycles else None
        # per solve (= per bench step) this kernel may run more or less than once (restart pass: once per step, on few QPs)
        kernels[short[k]] = kk
        if dom is None:
            dom = short[k]
    d0 = kernels[dom]
    calls0 = d0["rocprof_kernel_trace"]["calls"]
    for name, kk in kernels.items():
        w = kk["rocprof_kernel_trace"]["calls"] / calls0          # launches of this kernel per launch of the dominant one
        tot_us += w * kk["rocprof_kernel_trace"]["avg_us"]
        tot_fetch += w * kk["hbm_bytes_per_launch"]["fetch_x2"]; tot_write += w * kk["hbm_bytes_per_launch"]["write"]
        tot_flops += w * kk["executed_flops_per_launch"]["total"]
    ws = d0["wave_time_split"]
    vu, mu, lu = d0["valu_util"] or 0.0, d0["mfma_util"] or 0.0, d0["lds_busy"] or 0.0
    hbm_frac = (tot_fetch + tot_write) / (tot_us * 1e-6) / 8e12
    if hbm_frac > 0.5:
        bound = "hbm"
    elif mu > 0.6:
        bound = "mfma"
    elif vu > 0.6:
        bound = "valu-issue"
    elif lu > 0.6:
        bound = "lds"
    else:
        bound = "latency (valu %.0f %%, mfma %.0f %%, lds %.0f %% busy; waves issue %.0f %%, issue-stalled %.0f %%, parked on s_waitcnt / barriers %.0f %% of their time)" % (
            100 * vu, 100 * mu, 100 * lu, 100 * ws["issuing"], 100 * ws["issue_stalled"], 100 * ws["parked_waitcnt_or_barrier"])
    # which build the counters were taken on: bench.py compares library_sha256 with the library it has loaded and says so in its line
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libp = os.environ.get("SRBDQP_LIB") or os.path.join(root, "g1_locomotion_amd", "libsrbdqp.so")
    try:
        lib_sha = hashlib.sha256(open(libp, "rb").read()).hexdigest()
    except OSError:
        lib_sha = None
    try:
        git_rev = open(os.path.join(root, "g1_locomotion_amd", "libsrbdqp.rev")).read().strip()      # written by the build (tools/build.sh, __graft_entry__.build)
    except OSError:
        git_rev = None
    d = {"round": rnd, "bench_kernel_name": kname, "batch_per_launch": B, "horizon": N, "library_sha256": lib_sha, "git_rev": git_rev,
         "dominant_kernel": d0["rocprof_kernel_trace"]["name"],
         "workload": os.environ.get("PMC_WORKLOAD", "bench.py --streams 1 ...: one solve at a time; see `command`"),
         "command": "tools/pmc_collect.sh (one rocprofv3 --pmc <group> --kernel-trace pass per counter group) + rocprofv3 --kernel-trace --stats on the same bench command",
         "kernel_avg_us_rocprof_kernel_trace": tot_us,
         "hbm": {"fetch_bytes_corrected_x2": tot_fetch, "write_bytes": tot_write, "traffic_bytes_per_launch": tot_fetch + tot_write,
                 "algorithmic_bytes_per_launch": alg,
                 "traffic_bytes_per_solve": (tot_fetch + tot_write) * lps,
                 "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled per MI355X_MICROARCH.md (HBM section); WRITE_SIZE taken as is; "
                         "per launch = the mean over the launches of the dominant kernel (with the rho restart on, a solve is two launches of "
                         "it: the full pass and the pass over the capped QPs); per solve = x launches_per_solve"},
         "launches_per_solve": lps,
         "executed": {"flops_per_launch": tot_flops, "flops_per_solve": tot_flops * lps, "peak_TFLOPs": PEAK[esz],
                      "TFLOPs_at_rocprof_avg": tot_flops / (tot_us * 1e-6) / 1e12, "frac_at_rocprof_avg": tot_flops / (tot_us * 1e-6) / 1e12 / PEAK[esz],
                      "bound": bound,
                      "utilisation": {"valu_busy": vu, "mfma_busy": mu, "lds_busy": lu, "hbm_frac_of_8TBps": hbm_frac, "wave_time_split": ws,
                                      "lds_bank_conflict_share": d0["lds_bank_conflict_share"], "clock_GHz": d0["clock_GHz"]},
                      "note": "issued flops from the PMC counters: SQ_INSTS_VALU_FLOPS_FP32/FP64 x 64 lanes (= (FMA x 2 + ADD + MUL) per wave instruction, "
                              "packed fp32 instructions counted twice; EXEC-masked lanes counted) + MFMA x 2048; the useful share is lower (masked lanes, padded tiles)"},
         "kernels": kernels,
         "notes": "SQ_* are summed over the chip; SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES/SQ_WAIT_* are in 4-cycle quads; GRBM_GUI_ACTIVE is summed over the 8 XCDs."}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps({"avg_us_sum": tot_us, "hbm": d["hbm"], "executed": d["executed"]}, indent=1))


if __name__ == "__main__":
    main()
