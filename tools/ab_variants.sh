#!/bin/bash
# A/B on ONE GPU box (MI355X devices differ by a few per cent between boxes): builds library variants with extra
# -D flags (on top of gridworld_amd/build.py's FLAGS) and benches each in turn, REPS times, interleaved.
#   usage: tools/ab_variants.sh "" "-DIGW_AB_X" ...      (MODE=walking|flying, WORKLOAD=rt20|cdm, REPS=3, STEPS=400)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
FLAGS=$(python3 -c "from gridworld_amd import build; print(' '.join(build.FLAGS))")
i=0
for FL in "$@"; do
  OUT=gridworld_amd/libigw_ab$i.so
  if [ ! -f $OUT ] || [ gridworld_amd/csrc/igw_kernels.hip -nt $OUT ]; then
    /opt/rocm/bin/hipcc $FLAGS $FL -o $OUT gridworld_amd/csrc/igw_kernels.hip 2> gpurun_out/ab/build$i.log || tail -5 gpurun_out/ab/build$i.log
  fi
  i=$((i+1))
done
for rep in $(seq 1 ${REPS:-3}); do
  i=0
  for FL in "$@"; do
    IGW_LIB=$PWD/gridworld_amd/libigw_ab$i.so python3 bench.py --no-cpu-baseline --no-fused --no-async --no-secondary --no-api --mode ${MODE:-walking} --workload ${WORKLOAD:-rt20} --steps ${STEPS:-400} --warmup 20 --windows 3 --rehearsals 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('rep $rep variant $i [$FL] ${MODE:-walking} ${WORKLOAD:-rt20} kernel %.3f us  %.3f G  %s' % (d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9, d['config']['windows_kernel_us']))"
    i=$((i+1))
  done
done
