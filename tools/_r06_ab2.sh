export TMPDIR=/tmp
( for wl in "walking rt20" "walking cdm" "flying rt20"; do set -- $wl; MODE=$1 WORKLOAD=$2 REPS=3 IGW_AB_NO_STEP_COUNTER=1 tools/ab_libs.sh gridworld_amd/libigw_ab2.so gridworld_amd/libigw_ab5.so; done ) > gpurun_out/r06_ab_split.txt 2>&1
IGW_LIB=$PWD/gridworld_amd/libigw_ab5.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flying.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r06_split_parity.txt 2>&1
tail -30 gpurun_out/r06_ab_split.txt; tail -5 gpurun_out/r06_split_parity.txt
