#!/bin/bash
# Diagnostic (GPU box): kernel time of the walking step kernel with parts switched off (IGW_DIAG switches).
set -u
export IGW_DIAG=1
for F in ${FLAGS:-0 16 32 48 1 2 4 7 55 128 64}; do
  python3 bench.py --no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 3 --rehearsals 1 --mode ${MODE:-walking} --steps 300 --warmup 20 --debug-flags $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags $F kernel %.2f us' % (d['roofline']['kernel_avg_ms']*1e3))"
done

