#!/bin/bash
# A/B of prebuilt library variants on ONE GPU box: tools/ab_libs.sh lib1.so lib2.so ...   (MODE=walking|flying, REPS=3)
set -u
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-3}); do
  for LIB in "$@"; do
    IGW_LIB=$PWD/$LIB python3 bench.py --no-cpu-baseline --no-fused --no-async --no-secondary --no-api --mode ${MODE:-walking} --workload ${WORKLOAD:-rt20} --steps ${STEPS:-400} --warmup 20 --windows 3 --rehearsals 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('rep $rep %-34s ${MODE:-walking} ${WORKLOAD:-rt20} kernel %.3f us  %.3f G  windows %s' % ('$LIB', d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9, d['config']['windows_kernel_us']))"
  done
done
