#!/bin/bash
# A/B on ONE GPU box (MI355X devices differ by a few per cent between boxes): builds library variants with extra
# -D flags and benches each in turn, twice, interleaved.   usage: tools/ab_variants.sh "" "-DIGW_AB_X" ...
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
i=0
for FL in "$@"; do
  OUT=gridworld_amd/libigw_ab$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $FL -o $OUT gridworld_amd/csrc/igw_kernels.hip 2> gpurun_out/ab/build$i.log || cat gpurun_out/ab/build$i.log | tail -5
  i=$((i+1))
done
for rep in 1 2 3; do
  i=0
  for FL in "$@"; do
    IGW_LIB=$PWD/gridworld_amd/libigw_ab$i.so python3 bench.py --no-cpu-baseline --no-fused --no-async --mode ${MODE:-walking} --steps 400 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep variant $i [$FL] kernel %.3f us  %.3f G' % (d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9))"
    i=$((i+1))
  done
done
