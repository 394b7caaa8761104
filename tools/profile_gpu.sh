#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM traffic.
# Outputs land in gpurun_out/prof_<tag>/ ; summarise with tools/summarize_profile.py.
set -u
TAG=${1:-r01}
ARGS=${2:-"--no-cpu-baseline --no-fused --steps 250 --warmup 20"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $ARGS > $OUT/kt_bench.json 2> $OUT/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write_bench.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
# keep only the small summaries (kernel stats + per-dispatch counters of the step kernel)
python3 tools/summarize_profile.py $OUT $TAG
