export TMPDIR=/tmp IGW_AB_NO_STEP_COUNTER=1
run() { IGW_LIB=$PWD/gridworld_amd/libigw_$1.so python3 bench.py --no-cpu-baseline --no-fused --no-secondary --no-api --envs-per-gpu $2 --steps 400 --warmup 20 --windows 5 --rehearsals 1 --sweep "" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('rep $3 %-6s N=$2 lanes %d kernel %.3f us  %.4f G  %s' % ('$1', d['config']['lanes_per_env'], d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9, d['config']['windows_kernel_us']))"; }
( for n in 4096 1024 16384; do for rep in 1 2 3; do run ab30 $n $rep; run ab31 $n $rep; done; done ) > gpurun_out/r06_ab_smallorder.txt 2>&1
cat gpurun_out/r06_ab_smallorder.txt
