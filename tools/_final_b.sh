set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_config.sh r03 1 wave_f64_n10_s2 4096 10 8 1 > gpurun_out/profile_c1.log 2>&1 || echo "FAILED c1"
bash tools/profile_config.sh r03 2 wrench_f32_n20 65536 20 4 1 > gpurun_out/profile_c2.log 2>&1 || echo "FAILED c2"
bash tools/profile_config.sh r03 3 wave_f64_n10_s2 65536 10 8 1 > gpurun_out/profile_c3.log 2>&1 || echo "FAILED c3"
bash tools/profile_config.sh r03 4 ragged_wrench_f64_n8_n12_n16_n24 16384 15 8 2 > gpurun_out/profile_c4.log 2>&1 || echo "FAILED c4"
bash tools/profile_config.sh r03 1 compact_f64_n10_s2 4096 10 8 1 --kernel compact > gpurun_out/profile_c1c.log 2>&1 || echo "FAILED c1 compact"
bash tools/trace_two_streams.sh r03 > gpurun_out/trace2.log 2>&1 || echo "FAILED trace2"
bash tools/trace_ragged.sh r03 > gpurun_out/trace_ragged.log 2>&1 || echo "FAILED trace ragged"
tail -3 gpurun_out/trace_r03/r03_two_streams_timeline.txt; cat gpurun_out/trace_ragged_r03/r03_ragged_timeline.txt
