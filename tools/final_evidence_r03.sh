#!/bin/bash
# One gpurun call: everything DESIGN.md quotes for round 3.  Outputs under gpurun_out/r03_final/ and
# gpurun_out/profiles_r03*/ (copied into profiles/ by hand afterwards).
set -u
export TMPDIR=/tmp
D=gpurun_out/r03_final; mkdir -p $D
timeout 2400 python3 -m pytest tests -m gpu -q > $D/pytest_gpu.txt 2>&1; tail -3 $D/pytest_gpu.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver.json 2> $D/bench_driver.err
timeout 900 python3 bench.py > $D/bench.json 2> $D/bench.err
timeout 600 python3 bench.py --mode flying --no-cpu-baseline > $D/bench_flying.json 2> $D/bench_flying.err
timeout 600 python3 bench.py --workload cdm --no-cpu-baseline > $D/bench_cdm.json 2> $D/bench_cdm.err
timeout 600 python3 bench.py --envs-per-gpu 524288 --steps 100 --no-cpu-baseline --no-secondary > $D/bench_524288.json 2> $D/bench_524288.err
IGW_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-fused --no-async > $D/bench_2ranks_shared_gpu.json 2> $D/bench_2ranks.err
timeout 300 python3 __graft_entry__.py smoke > $D/smoke.txt 2>&1; tail -1 $D/smoke.txt
timeout 1500 bash tools/profile_gpu.sh r03 > $D/profile_walk.txt 2>&1
timeout 1500 bash tools/profile_gpu.sh r03_flying "--mode flying --no-cpu-baseline --no-fused --no-async --no-secondary --windows 3 --rehearsals 1 --steps 200 --warmup 20" > $D/profile_fly.txt 2>&1
timeout 300 python3 tools/stamp_phases.py $D/stamps_nodrain.npz 4 8 > $D/phase_stamps_nodrain.txt 2>&1
timeout 300 python3 tools/stamp_phases.py $D/stamps.npz 4 > $D/phase_stamps.txt 2>&1
FLAGS="0 16 32 1 2 4 7 256 128 64" bash tools/ablate_time.sh 2>/dev/null | grep "^flags" > $D/ablation_time.txt
rm -f $D/*.npz
rm -rf gpurun_out/prof_r03 gpurun_out/prof_r03_flying   # raw counter dumps: tens of MB; the summaries are in gpurun_out/profiles_r03*
for f in bench_driver bench bench_flying bench_cdm bench_524288 bench_2ranks_shared_gpu; do
  python3 - $D/$f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1] + '.json').read().strip().splitlines() if l.startswith('{')][-1])
    c = d['config']
    print('%-26s %.3f G  ms/step %.5f  kernel %.2f us  frac %.3f design %.3f  spread %.3f  p %.4f  %s | %s' % (
        sys.argv[1].split('/')[-1], d['value'] / 1e9, d['ms_per_step'], 1e3 * d['roofline']['kernel_avg_ms'], d['roofline']['frac'],
        d['roofline']['frac_design'], c['window_spread'], c['p_changed'], c['timed_as'], c['step_count_gather']))
    for k in ('flying', 'cdm'):
        if k in c:
            print('   %-8s %.3f G  kernel %.2f us  p %.4f cell %.4f resets %d' % (k, c[k]['value'] / 1e9, c[k]['kernel_us'], c[k]['p_changed'], c[k]['p_cell_changed'], c[k]['resets_in_window']))
    for k in ('fused_rollout_env_steps_per_s', 'fused_rollout_recorded_actions_env_steps_per_s', 'async_2_subbatches_env_steps_per_s'):
        if c.get(k): print('   %s %.3f G' % (k, c[k] / 1e9))
    if 'cpu_baseline' in d:
        b = d['cpu_baseline']
        print('   cpu: %.3f M on %d threads (affinity %s quota %s effective %.1f), 1 core %.3f M; flying %.3f M / %.3f M; config0 %.1f k' % (
            b['value'] / 1e6, b['cores'], b['affinity_cpus'], b['cgroup_cpu_quota'], b['effective_cores'], b['value_1core'] / 1e6,
            b['flying']['value'] / 1e6, b['flying']['value_1core'] / 1e6, b['config0']['value'] / 1e3))
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1] + '.err').read()[-1500:])
PY
done
timeout 900 python3 tests/fuzz_parity.py 100 2026 > $D/fuzz_100.txt 2>&1; tail -1 $D/fuzz_100.txt
tail -22 $D/phase_stamps_nodrain.txt | head -12; cat $D/ablation_time.txt
