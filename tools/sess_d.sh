set -u
export TMPDIR=/tmp
D=gpurun_out/r03d; mkdir -p $D
bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vDIGW_OBS_LATE1.so tools/ab/libigw_vDMAASM.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vDIGW_OBS_LATE1.so tools/ab/libigw_vDMAASM.so 2>&1 | tee $D/ab_fly.txt
IGW_LIB=$PWD/tools/ab/libigw_vDMAASM.so timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
