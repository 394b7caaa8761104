set -u
export TMPDIR=/tmp
REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vM.so tools/ab/libigw_vO.so 2>&1
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vM.so tools/ab/libigw_vO.so 2>&1
IGW_LIB=$PWD/tools/ab/libigw_vO.so timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
