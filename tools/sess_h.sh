set -u
export TMPDIR=/tmp
D=gpurun_out/r03p; mkdir -p $D
REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vF.so tools/ab/libigw_vG.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vF.so tools/ab/libigw_vG.so 2>&1 | tee $D/ab_fly.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_flying.py -x -q 2>&1 | tail -2
