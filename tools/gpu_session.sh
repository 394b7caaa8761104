#!/bin/bash
# One gpurun call = one session: parity tests, bench variants, phase stamps, profiles.  Logs in gpurun_out/<tag>/.
set -u
TAG=${1:-s1}
MODE=${2:-full}
D=gpurun_out/$TAG
mkdir -p $D
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1
tail -15 $D/pytest_gpu.txt
timeout 600 python3 bench.py > $D/bench_default.json 2> $D/bench_default.err
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $D/bench_20.json 2> $D/bench_20.err
timeout 300 python3 bench.py --mode flying --no-cpu-baseline > $D/bench_flying.json 2> $D/bench_flying.err
timeout 300 python3 tools/stamp_phases.py $D/stamps.npz 4 > $D/stamps.txt 2>&1
timeout 300 python3 tools/stamp_phases.py $D/stamps_nodrain.npz 4 8 > $D/stamps_nodrain.txt 2>&1
if [ "$MODE" = "full" ]; then
  timeout 300 python3 bench.py --envs-per-gpu 1048576 --steps 50 --no-cpu-baseline --no-fused > $D/bench_1m.json 2> $D/bench_1m.err
  timeout 300 python3 bench.py --lanes-per-env 2 --no-cpu-baseline --no-fused > $D/bench_gs2.json 2> $D/bench_gs2.err
  timeout 300 python3 bench.py --lanes-per-env 8 --no-cpu-baseline --no-fused > $D/bench_gs8.json 2> $D/bench_gs8.err
  timeout 1200 bash tools/profile_gpu.sh $TAG > $D/profile.txt 2>&1
fi
for f in bench_default bench_20 bench_flying bench_1m bench_gs2 bench_gs8; do
  [ -f $D/$f.json ] && python3 - $D/$f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1] + '.json').read().strip().splitlines()[-1])
    print('%-28s %.3f G  ms/step %.4f  kernel %.4f ms  frac %.3f  resets %s  p %.4f  fused %s  async2 %s' % (
        sys.argv[1].split('/')[-1], d['value'] / 1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'],
        d['config']['resets_in_window'], d['config']['p_changed'], d['config']['fused_rollout_env_steps_per_s'],
        d['config'].get('async_2_subbatches_env_steps_per_s')))
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1] + '.err').read()[-800:])
PY
done
head -12 $D/stamps.txt
head -24 $D/stamps_nodrain.txt
