set -u
export TMPDIR=/tmp
D=gpurun_out/r03h; mkdir -p $D
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_flying.py tests/test_gpu_facade.py tests/test_gpu_trajectory.py -x -q > $D/pytest.txt 2>&1
tail -4 $D/pytest.txt
bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vlate.so tools/ab/libigw_vsdwa.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vbase.so tools/ab/libigw_vlate.so tools/ab/libigw_vsdwa.so 2>&1 | tee $D/ab_fly.txt
