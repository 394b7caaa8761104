#!/bin/bash
# One gpurun call of round 3: parity tests, the driver's bench command, the 500-step bench, two ranks sharing the GPU
# (mixed gloo + RCCL group: exercises init, the thread-local capture and the RCCL-gather fallback), smoke.
set -u
TAG=${1:-r03a}
D=gpurun_out/$TAG
mkdir -p $D
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1
  tail -15 $D/pytest_gpu.txt
fi
IGW_BENCH_TRACE=1 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_20.json 2> $D/bench_20.err
timeout 600 python3 bench.py --no-cpu-baseline --no-secondary > $D/bench_500.json 2> $D/bench_500.err
IGW_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-fused --no-async > $D/bench_2ranks.json 2> $D/bench_2ranks.err
timeout 300 python3 __graft_entry__.py smoke > $D/smoke.txt 2>&1
tail -2 $D/smoke.txt
for f in bench_20 bench_500 bench_2ranks; do
  python3 - $D/$f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1] + '.json').read().strip().splitlines() if l.startswith('{')][-1])
    c = d['config']
    print('%-14s %.3f G  ms/step %.5f  kernel %.2f us  frac %.3f design %.3f  windows %s spread %.3f  rehearsals %s  p %.4f  %s | %s' % (
        sys.argv[1].split('/')[-1], d['value'] / 1e9, d['ms_per_step'], 1e3 * d['roofline']['kernel_avg_ms'], d['roofline']['frac'],
        d['roofline']['frac_design'], c['windows_ms_per_step'], c['window_spread'], c['rehearsal_ms_per_step'], c['p_changed'],
        c['timed_as'], c['step_count_gather']))
    for k in ('flying', 'cdm'):
        if k in c:
            print('   %-8s %.3f G  kernel %.2f us  windows %s  p %.4f cell %.4f' % (k, c[k]['value'] / 1e9, c[k]['kernel_us'], c[k]['windows_ms_per_step'], c[k]['p_changed'], c[k]['p_cell_changed']))
    if 'cpu_baseline' in d:
        b = d['cpu_baseline']
        print('   cpu: %.3f M on %d threads (affinity %s quota %s effective %.1f), 1 core %.3f M; flying %.3f M / %.3f M; config0 %.1f k' % (
            b['value'] / 1e6, b['cores'], b['affinity_cpus'], b['cgroup_cpu_quota'], b['effective_cores'], b['value_1core'] / 1e6,
            b['flying']['value'] / 1e6, b['flying']['value_1core'] / 1e6, b['config0']['value'] / 1e3))
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1] + '.err').read()[-1500:])
PY
done
grep "host us" $D/bench_20.err | tail -8
