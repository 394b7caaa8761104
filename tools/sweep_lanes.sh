#!/bin/bash
# Diagnostic (GPU box): env-steps/s over batch size x lanes per env, to (re)tune the automatic group width.
set -u
export TMPDIR=/tmp
for N in ${NS:-4096 16384 65536 131072 262144 1048576}; do
  for GS in ${LANES:-64 16 8 4 2 1}; do
    W=$((N * GS / 64))
    if [ $W -lt 512 ] || [ $W -gt 70000 ]; then continue; fi
    K=$((20000000 / N)); [ $K -gt 400 ] && K=400; [ $K -lt 30 ] && K=30
    python3 bench.py --no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 3 --rehearsals 1 --steps $K --warmup 10 --envs-per-gpu $N --lanes-per-env $GS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('N $N lanes $GS waves $W: %.3f G  kernel %.2f us' % (d['value']/1e9, d['roofline']['kernel_avg_ms']*1e3))"
  done
done
