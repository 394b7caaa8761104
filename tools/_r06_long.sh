export TMPDIR=/tmp
python3 tests/afly_divergence.py --action-space walking_dict --passes 7 --seed0 9100 --out gpurun_out/adict_divergence_r06.json > gpurun_out/adict_r06.log 2>&1
tail -30 gpurun_out/adict_r06.log | head -40
python3 tests/scenario_fuzz.py 12000 3000000 gpurun_out/fuzz_facade_vs_glibc_r06.json glibc > gpurun_out/fuzz_facade_glibc_r06.log 2>&1
tail -3 gpurun_out/fuzz_facade_glibc_r06.log
python3 tests/fuzz_parity.py 900 60600 > gpurun_out/fuzz_extended_r06.txt 2>&1
tail -3 gpurun_out/fuzz_extended_r06.txt
