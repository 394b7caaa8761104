#!/bin/bash
# GPU box: bench two prebuilt libraries against each other on one box (gridworld_amd/libigw_base.so = the previous
# commit's build, copied in by hand; gridworld_amd/libigw_hip.so = the working tree).  MODE=flying for configs[3].
for rep in 1 2 3; do for L in libigw_base.so libigw_hip.so; do
  IGW_LIB=$PWD/gridworld_amd/$L python3 bench.py --no-cpu-baseline --no-fused --no-async --mode ${MODE:-walking} --steps 400 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep $L kernel %.3f us %.3f G' % (d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9))"
done; done
