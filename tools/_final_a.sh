set -u
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r03 bench 2>&1 | tail -3
bash tools/refresh_profiles.sh r03 sweeps 2>&1 | tail -3
python tools/ragged_bench.py > gpurun_out/refresh/r03_ragged_config4.txt 2>/dev/null
python tools/latency_patterns.py 3000 2>/dev/null | sed -n '/^{/,$p' > gpurun_out/refresh/r03_latency_patterns.json
python tools/wrench_stamps.py 20 double 1 1 > gpurun_out/refresh/r03_wrench_f32_n20_phase_stamps.txt 2>/dev/null
python tools/wrench_stamps.py 20 double 65536 1 >> gpurun_out/refresh/r03_wrench_f32_n20_phase_stamps.txt 2>/dev/null
SCHED=double python tools/wrench_stamps_staged.py > gpurun_out/refresh/r03_wrench_f64_n10_lat_staged_stamps.txt 2>/dev/null
ls -la gpurun_out/refresh | grep r03_ | wc -l
