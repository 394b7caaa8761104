set -u
export TMPDIR=/tmp
D=gpurun_out/r03f; mkdir -p $D
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_flying.py tests/test_gpu_random_tasks.py -x -q > $D/pytest.txt 2>&1
tail -5 $D/pytest.txt
bash tools/ab_libs.sh tools/ab/libigw_vbase.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_walk.txt
MODE=flying REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vbase.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_fly.txt
WORKLOAD=cdm REPS=2 bash tools/ab_libs.sh tools/ab/libigw_vbase.so gridworld_amd/libigw_hip.so 2>&1 | tee $D/ab_cdm.txt
