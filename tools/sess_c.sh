set -u
export TMPDIR=/tmp
D=gpurun_out/r03c; mkdir -p $D
timeout 300 python3 tools/stamp_phases.py $D/stamps_nodrain.npz 4 8 > $D/stamps_nodrain.txt 2>&1
cat $D/stamps_nodrain.txt | tail -42
FLAGS="0 1 2 4 16 256" bash tools/ablate_time.sh 2>&1 | tee $D/ablate.txt
