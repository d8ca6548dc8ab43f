#!/bin/bash
# A/B of two builds of the library over a list of tools/schedule_bench.py cases, on the GPU box (through gpurun):
#   tools/ab_builds.sh <library B (.so, inside the repo so that it travels)> <out file> [cases file, one schedule_bench argument list per line]
# A = the in-tree g1_locomotion_amd/libsrbdqp.so; B is loaded through SRBDQP_LIB.  Build B with e.g.
#   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Iinclude -D<switch> -o tools/_ab/lib_b.so g1_locomotion_amd/csrc/srbdqp.hip
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
LB=$(realpath "$1"); O=$2; CASES=${3:-}
: > "$O"
run_cases() {
  while read -r args; do
    [ -z "$args" ] && continue
    for v in A B; do
      if [ $v = A ]; then unset SRBDQP_LIB; else export SRBDQP_LIB=$LB; fi
      echo -n "$v " >> "$O"
      timeout -k 10 200 python "$R/tools/schedule_bench.py" $args 2>&1 | tail -1 >> "$O"
    done
  done
}
if [ -n "$CASES" ]; then run_cases < "$CASES"; else run_cases <<DEFAULT
mixed 10 4096 auto 0
double 10 4096 auto 0
mixed 8 4096 auto 0
mixed 12 16384 auto 0
double 16 16384 auto 0
double 20 16384 auto 0
mixed 24 16384 auto 0
mixed 10 4096 auto 1
mixed 12 16384 auto 1
mixed 16 16384 auto 1
mixed 24 16384 auto 1
DEFAULT
fi
