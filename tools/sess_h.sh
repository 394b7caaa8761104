set -u
export TMPDIR=/tmp
D=gpurun_out/r03m; mkdir -p $D
REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vD.so tools/ab/libigw_vE.so 2>&1 | tee $D/ab_walk.txt
WORKLOAD=cdm REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vD.so tools/ab/libigw_vE.so 2>&1 | tee $D/ab_cdm.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
