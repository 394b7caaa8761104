#!/bin/bash
# GPU box, one call = one box: fused-rollout throughput of library variants built with extra -D flags.
# usage: tools/rollout_ab.sh "" "-DIGW_AB_ROLLOUT_LB=3" ...
set -u
i=0
for FL in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $FL -o gridworld_amd/libigw_ab$i.so gridworld_amd/csrc/igw_kernels.hip 2>/dev/null
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for FL in "$@"; do
    echo "rep $rep variant $i [$FL] $(IGW_LIB=$PWD/gridworld_amd/libigw_ab$i.so python3 tools/rollout_actions_bench.py 2>/dev/null | tr '\n' ' ' | sed 's/  */ /g')"
    i=$((i+1))
  done
done
