#!/bin/bash
# Round 6, same-box A/B of the step counter and the block size (run through gpurun; the variant libraries are built
# in the container first: tools/ab_r06_counter.sh build).  Variants:
#   0 steps counter off (= the round-4 kernel)   1 once per block (= round 5)   2 once per launch (HEAD)
#   3 once per launch + 512-thread blocks        4 once per block + 512-thread blocks
set -u
export TMPDIR=/tmp
VARS=("-DIGW_STEPS_MODE=0" "-DIGW_STEPS_MODE=1" "-DIGW_STEPS_MODE=2" "-DIGW_STEPS_MODE=2 -DIGW_BLOCK=512" "-DIGW_STEPS_MODE=1 -DIGW_BLOCK=512")
if [ "${1:-}" = build ]; then
  FLAGS=$(python3 -c "from gridworld_amd import build; print(' '.join(build.FLAGS))")
  for i in 0 1 2 3 4; do
    /opt/rocm/bin/hipcc $FLAGS ${VARS[$i]} -DIGW_BUILD_ID="\"igw-build-id:ab$i\"" -o gridworld_amd/libigw_ab$i.so gridworld_amd/csrc/igw_kernels.hip &
  done
  wait; ls -la gridworld_amd/libigw_ab*.so; exit 0
fi
for wl in "walking rt20" "walking cdm" "flying rt20"; do
  set -- $wl
  for rep in 1 2 3; do
    for i in 0 1 2 3 4; do
      IGW_AB_NO_STEP_COUNTER=1 IGW_LIB=$PWD/gridworld_amd/libigw_ab$i.so python3 bench.py --no-cpu-baseline --no-fused --no-secondary --no-api --mode $1 --workload $2 --steps ${STEPS:-400} --warmup 20 --windows 3 --rehearsals 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('rep $rep variant $i [${VARS[$i]}] $1 $2 kernel %.3f us  %.3f G  %s' % (d['roofline']['kernel_avg_ms']*1e3, d['value']/1e9, d['config']['windows_kernel_us']))"
    done
  done
done
