#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace of a short (driver-sized) bench window, then prints the start
# offset, duration and gap of the last K step-kernel dispatches -- where a 20-step window spends its time.
set -u
K=${1:-20}
OUT=gpurun_out/trace_window
mkdir -p $OUT
export TMPDIR=/tmp
for mode in graph nograph; do
    extra=""; [ $mode = nograph ] && extra="--no-graph"
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -- python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-fused --no-async $extra > $OUT/$mode.json 2> $OUT/$mode.log
    python3 - $OUT/$mode $K <<'PY'
import csv, glob, sys
d, K = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'][:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
idx = [i for i, k in enumerate(ks) if 'step_kernel' in k[0]]
last = idx[-K:]
t0 = ks[last[0]][1]
print(d, 'last', K, 'step kernels; all kernels from 8 before the window:')
prev_end = None
for i in range(last[0] - 8, last[-1] + 1):
    n, s, e = ks[i]
    print('  %-40s start %9.1f us  dur %6.2f us  gap %7.2f us' % (n, (s - t0) / 1e3, (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
print('  window: first start -> last end %.1f us; sum of durations %.1f us' % ((ks[last[-1]][2] - t0) / 1e3, sum(ks[i][2] - ks[i][1] for i in last) / 1e3))
PY
    cat $OUT/$mode.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench: %.3f G ms/step %.4f kernel %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
