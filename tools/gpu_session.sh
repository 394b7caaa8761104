#!/bin/bash
# One gpurun call = one session: sanity tests, bench variants, phase stamps, profiles.  Logs in gpurun_out/<tag>/.
set -u
TAG=${1:-s1}
D=gpurun_out/$TAG
mkdir -p $D
export TMPDIR=/tmp
python3 -c "import torch; print(torch.cuda.get_device_name(0))" > $D/device.txt 2>&1
timeout 600 python3 bench.py > $D/bench_default.json 2> $D/bench_default.err
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $D/bench_20.json 2> $D/bench_20.err
timeout 300 python3 bench.py --no-graph --no-cpu-baseline > $D/bench_nograph.json 2> $D/bench_nograph.err
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $D/bench_20_nograph.json 2> $D/bench_20_nograph.err
timeout 300 python3 bench.py --lockstep --no-graph --no-cpu-baseline > $D/bench_lockstep.json 2> $D/bench_lockstep.err
timeout 300 python3 bench.py --mode flying --no-cpu-baseline > $D/bench_flying.json 2> $D/bench_flying.err
timeout 300 python3 tools/stamp_phases.py $D/stamps.npz 4 > $D/stamps.txt 2>&1
timeout 1200 bash tools/profile_gpu.sh $TAG > $D/profile.txt 2>&1
timeout 900 python3 -m pytest tests -m gpu -x -q > $D/pytest_gpu.txt 2>&1
tail -3 $D/pytest_gpu.txt
cat $D/bench_default.json | head -c 3000
