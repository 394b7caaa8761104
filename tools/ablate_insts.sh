#!/bin/bash
# Diagnostic (GPU box): dynamic instruction counts of the walking step kernel with parts of the step switched off
# (IGW_DIAG ablation switches: 1 = no histogram update, 2 = no sight / ray march, 4 = no physics sub-steps).
set -u
OUT=gpurun_out/ablate
mkdir -p $OUT
export TMPDIR=/tmp IGW_DIAG=1
for F in ${FLAGS:-0 1 2 4 6 7}; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/f$F -- python3 bench.py --no-cpu-baseline --no-fused --no-async --no-secondary --no-api --windows 2 --rehearsals 0 --mode ${MODE:-walking} --steps 100 --warmup 10 --debug-flags $F > $OUT/f$F.json 2> $OUT/f$F.log
done
export FLAGS="${FLAGS:-0 1 2 4 6 7}"
python3 - <<'PY'
import csv, glob, collections
import os
for F in [int(x) for x in os.environ.get('FLAGS', '0 1 2 4 6 7').split()]:
    f = glob.glob(f'gpurun_out/ablate/f{F}/**/*counter_collection.csv', recursive=True)
    if not f:
        print(F, 'no data'); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if 'step_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    w = acc['SQ_WAVES'][1] / max(acc['SQ_WAVES'][0], 1)
    print('flags', F, {k: round(s / n / w, 1) for k, (n, s) in sorted(acc.items()) if k != 'SQ_WAVES'})
PY
rm -rf gpurun_out/ablate/f*/
