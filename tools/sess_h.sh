set -u
export TMPDIR=/tmp
D=gpurun_out/r03n; mkdir -p $D
REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vD.so tools/ab/libigw_vF.so 2>&1 | tee $D/ab_walk.txt
WORKLOAD=cdm REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vD.so tools/ab/libigw_vF.so 2>&1 | tee $D/ab_cdm.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
timeout 600 python3 tests/fuzz_parity.py 60 77 2>&1 | tail -2
