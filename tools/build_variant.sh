#!/bin/bash
# diagnostic builds: tools/build_variant.sh -DSRBDQP_PROFILE_ADMM [-D...]   (overwrites libsrbdqp.so; rebuild with tools/build.sh afterwards)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include "$@" -o $R/g1_locomotion_amd/libsrbdqp.so $R/g1_locomotion_amd/csrc/srbdqp.hip -lhsa-runtime64
