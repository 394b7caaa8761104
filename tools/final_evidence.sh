#!/bin/bash
# GPU box: the evidence DESIGN.md quotes beyond the standard session -- instruction / time ablations of the
# diagnostic build, the batch-size x lanes sweep.  Outputs under gpurun_out/evidence/ (copy into profiles/).
set -u
mkdir -p gpurun_out/evidence
bash tools/ablate_insts.sh 2>/dev/null | grep "^flags" > gpurun_out/evidence/ablation_insts.txt
bash tools/ablate_time.sh 2>/dev/null | grep "^flags" > gpurun_out/evidence/ablation_time.txt
bash tools/sweep_lanes.sh 2>/dev/null | grep "^N " > gpurun_out/evidence/sweep_lanes.txt
cat gpurun_out/evidence/*.txt
