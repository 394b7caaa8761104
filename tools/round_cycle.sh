#!/bin/bash
# The end-of-round measurement cycle on ONE GPU box (run through gpurun from the repo root; about 50 minutes):
#   1. rocprofv3 profiles (kernel trace + separate PMC passes, tools/profile_gpu.sh) of the four bench workloads -- walking
#      configs[2], flying configs[3], CDM targets, and the 2,097,152-env batch -- summarised into gpurun_out/profiles_<tag>/
#      and copied into profiles/ (so that the bench lines below quote profiles of the library they time);
#   2. the bench lines: the driver's command (--steps 20 --warmup 5, timed), the default 500-step run, flying, CDM, and the
#      64-lanes-per-env instantiation;
#   3. pytest -m gpu.
# Set IGW_GIT_COMMIT below to the commit of the tree (the GPU box has no .git); copy gpurun_out/bench_*.json and the pytest
# tail into profiles/ afterwards.
export TMPDIR=/tmp IGW_GIT_COMMIT=61ece0f
A="--no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 3 --rehearsals 1 --steps 200 --warmup 20"
tools/profile_gpu.sh r06 "$A" > gpurun_out/profile_r06.log 2>&1
tools/profile_gpu.sh r06_flying "$A --mode flying" > gpurun_out/profile_r06_flying.log 2>&1
tools/profile_gpu.sh r06_cdm "$A --workload cdm" > gpurun_out/profile_r06_cdm.log 2>&1
tools/profile_gpu.sh r06_large "--no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 2 --rehearsals 1 --steps 20 --warmup 5 --envs-per-gpu 2097152" > gpurun_out/profile_r06_large.log 2>&1
ls gpurun_out/profiles_r06*/
for t in r06 r06_flying r06_cdm r06_large; do cp gpurun_out/profiles_$t/${t}_*.json gpurun_out/profiles_$t/${t}_kernel_stats.csv profiles/; done
( time python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_20steps.json 2> gpurun_out/bench_r06_20.err ) 2> gpurun_out/bench_r06_20.time
cat gpurun_out/bench_r06_20.time
python3 bench.py --lanes-per-env 64 --no-secondary --no-cpu-baseline --no-api --no-fused --steps 100 --windows 5 --sweep '' > gpurun_out/bench_r06_lanes64.json 2> gpurun_out/bench_r06_lanes64.err
python3 bench.py > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
python3 bench.py --mode flying --no-cpu-baseline > gpurun_out/bench_r06_flying.json 2> gpurun_out/bench_r06_flying.err
python3 bench.py --workload cdm --no-cpu-baseline > gpurun_out/bench_r06_cdm.json 2> gpurun_out/bench_r06_cdm.err
python3 -c "
import json
for f in ('bench_r06_20steps','bench_r06','bench_r06_flying','bench_r06_cdm','bench_r06_lanes64'):
    d=json.loads([l for l in open('gpurun_out/'+f+'.json').read().splitlines() if l.startswith('{')][-1])
    print(f, '%.3f G' % (d['value']/1e9), 'ms/step %.5f' % d['ms_per_step'], 'kernel %.3f us' % (d['roofline']['kernel_avg_ms']*1e3), 'spread', d['config']['window_spread'], d['config']['window_spread_p10_p90'])
"
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r06a.txt 2>&1
tail -6 gpurun_out/pytest_r06a.txt
