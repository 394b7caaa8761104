set -u
export TMPDIR=/tmp
D=gpurun_out/r03k; mkdir -p $D
REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vA.so tools/ab/libigw_vD.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vA.so tools/ab/libigw_vD.so 2>&1 | tee $D/ab_fly.txt
WORKLOAD=cdm REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vA.so tools/ab/libigw_vD.so 2>&1 | tee $D/ab_cdm.txt
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
