#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats, separate PMC passes for HBM traffic and
# for the SQ issue-side counters.  Raw output lands in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py
# condenses it into gpurun_out/profiles_<tag>/<tag>_{kernel_stats.csv,traffic.json,issue.json}
# (copy those into profiles/ to commit them).  Call it as  IGW_GIT_COMMIT=<rev> tools/profile_gpu.sh <tag> "<bench args>":
# the GPU box has no .git, the summaries are stamped with that commit and with the library's build id.
set -u
TAG=${1:-r05}
ARGS=${2:-"--no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 3 --rehearsals 1 --steps 200 --warmup 20"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, rocprofv3 options...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/${name}_bench.json 2> $OUT/$name.log
}
run kt --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq_insts --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_cycles --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq_misc --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
python3 tools/summarize_profile.py $OUT $TAG
# the raw rocprofv3 trees are tens of MB per workload (gpurun merges at most 64 MiB back): keep them only on request
if [ -z "${KEEP_RAW:-}" ]; then rm -rf $OUT/*/; fi
