#!/bin/bash
# rebuild libsrbdqp.so (gfx950) and print the resource usage of the N=10 kernels
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -o $R/g1_locomotion_amd/libsrbdqp.so \
    $R/g1_locomotion_amd/csrc/srbdqp.hip -lhsa-runtime64 -Rpass-analysis=kernel-resource-usage 2> /tmp/srbdqp_build.log || { grep -E "error" -A6 /tmp/srbdqp_build.log | head -60; exit 1; }
( cd $R && echo "$(git rev-parse HEAD 2>/dev/null || echo unknown)$(git diff --quiet HEAD -- g1_locomotion_amd/csrc include 2>/dev/null || echo '+uncommitted')" > $R/g1_locomotion_amd/libsrbdqp.rev )
grep -E "warning" -A3 /tmp/srbdqp_build.log | head -20 || true
grep -A9 "Function Name: .*ILi10E" /tmp/srbdqp_build.log | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//'
