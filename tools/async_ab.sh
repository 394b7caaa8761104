#!/bin/bash
# GPU box, one call = one box: throughput of 1 / 2 sub-batches (tools/async_split.py) with different issue-priority
# maps (compile-time).   usage: tools/async_ab.sh "" "-DIGW_PRIO_MAP=0x33221100" "-DIGW_PRIO_MAP=0" ...
set -u
i=0
for FL in "$@"; do
  OUT=gridworld_amd/libigw_ab$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $FL -o $OUT gridworld_amd/csrc/igw_kernels.hip 2>/dev/null
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for FL in "$@"; do
    echo "rep $rep variant $i [$FL] $(IGW_LIB=$PWD/gridworld_amd/libigw_ab$i.so python3 tools/async_split.py 2>/dev/null | grep 'parts [12]:' | sed 's/ env-steps.*//' | tr '\n' ' ')"
    i=$((i+1))
  done
done
