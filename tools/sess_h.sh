set -u
export TMPDIR=/tmp
MODE=flying REPS=3 bash tools/ab_libs.sh tools/ab/libigw_vG.so tools/ab/libigw_vH.so 2>&1
timeout 900 python3 -m pytest tests/test_gpu_flying.py tests/test_gpu_fuzz.py tests/test_gpu_facade.py -x -q 2>&1 | tail -2
