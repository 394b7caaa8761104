#!/bin/bash
# GPU box: ms/step of every untimed rehearsal pass and of the measured pass of a driver-sized window, with the
# fraction of grid-changing steps in the measured pass (a falling curve with a falling p is workload drift from
# replaying the same actions, not a warm-up).
for R in ${RS:-3 30}; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused --no-async --rehearsals $R 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('R=$R: %.3f G ms/step %.4f kernel %.4f p_changed %.4f resets %d' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms'], c['p_changed'], c['resets_in_window'])); print(' '.join('%.2f' % (1e3*x) for x in c['rehearsal_ms_per_step']))"
done
