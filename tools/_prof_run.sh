#!/bin/bash
cd /root/repo
bash tools/profile_config.sh r03 1 wave_f64_n10_s2 4096 10 8 1 > gpurun_out/profile_c1.log 2>&1 || echo "FAILED profile c1"
bash tools/profile_config.sh r03 3 wave_f64_n10_s2 65536 10 8 1 > gpurun_out/profile_c3.log 2>&1 || echo "FAILED profile c3"
bash tools/trace_two_streams.sh r03 > gpurun_out/trace2.log 2>&1 || echo "FAILED trace"
bash tools/refresh_profiles.sh r03 sweeps > gpurun_out/sweeps.log 2>&1 || echo "FAILED sweeps"
ls gpurun_out/prof_r03_c1 gpurun_out/prof_r03_c3 gpurun_out/trace_r03 | head -40
