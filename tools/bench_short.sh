#!/bin/bash
# Runs on the GPU box: the driver-sized window (20 steps) three times, graph and eager, then the default run.
show() {
python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$1: %.3f G ms/step %.4f kernel %.4f resets %s rehearsals %s' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms'], c['resets_in_window'], c.get('rehearsal_ms_per_step')))"
}
for i in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-async 2>/dev/null | show "20-step"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-graph 2>/dev/null | show "20-step eager"
done
python3 bench.py --no-cpu-baseline --no-fused 2>/dev/null | show "500-step"
python3 bench.py --mode flying --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "flying 20-step"
