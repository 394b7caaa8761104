set -u
export TMPDIR=/tmp
D=gpurun_out/r03b; mkdir -p $D
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random_tasks.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_integration_doc.py tests/test_gpu_trajectory.py -x -q > $D/pytest.txt 2>&1
tail -12 $D/pytest.txt
bash tools/ab_libs.sh tools/ab/libigw_prev.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_prev.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_fly.txt
WORKLOAD=cdm REPS=2 bash tools/ab_libs.sh tools/ab/libigw_prev.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_cdm.txt
IGW_BENCH_TRACE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $D/bench_20.json 2> $D/bench_20.err
python3 -c "
import json
d=json.loads([l for l in open('$D/bench_20.json').read().splitlines() if l.startswith('{')][-1]); c=d['config']
print('20-step: %.3f G ms/step %.5f kernel %.2f windows %s %s' % (d['value']/1e9, d['ms_per_step'], 1e3*d['roofline']['kernel_avg_ms'], c['windows_ms_per_step'], c['timed_as']))"
grep "host us" $D/bench_20.err | tail -4
