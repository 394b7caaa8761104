#!/bin/bash
# Runs on the GPU box: host-side timeline of the timed window of a driver-sized bench run.
export IGW_BENCH_TRACE=1
for i in 1 2 3; do
  for extra in "" "--no-graph"; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused $extra 2>&1 >/dev/null | grep "host us" | sed "s/^/[$extra] /"
  done
done
python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-fused 2>&1 >/dev/null | grep "host us"
