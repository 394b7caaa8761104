export TMPDIR=/tmp IGW_GIT_COMMIT=def4827
A="--no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 3 --rehearsals 1 --steps 200 --warmup 20"
tools/profile_gpu.sh r06 "$A" > gpurun_out/profile_r06.log 2>&1
tools/profile_gpu.sh r06_flying "$A --mode flying" > gpurun_out/profile_r06_flying.log 2>&1
tools/profile_gpu.sh r06_cdm "$A --workload cdm" > gpurun_out/profile_r06_cdm.log 2>&1
tools/profile_gpu.sh r06_large "--no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 2 --rehearsals 1 --steps 20 --warmup 5 --envs-per-gpu 2097152" > gpurun_out/profile_r06_large.log 2>&1
ls gpurun_out/profiles_r06*/
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06a_20.json 2> gpurun_out/bench_r06a_20.err
tail -c 3000 gpurun_out/bench_r06a_20.json; tail -5 gpurun_out/bench_r06a_20.err
