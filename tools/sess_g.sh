set -u
export TMPDIR=/tmp
D=gpurun_out/r03i; mkdir -p $D
bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vA.so tools/ab/libigw_vB.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vA.so tools/ab/libigw_vB.so 2>&1 | tee $D/ab_fly.txt
WORKLOAD=cdm REPS=1 bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vA.so tools/ab/libigw_vB.so 2>&1 | tee $D/ab_cdm.txt
