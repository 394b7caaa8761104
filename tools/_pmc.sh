export TMPDIR=/tmp
for g in 4 1; do
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  d=gpurun_out/pmc_${g}_$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 bench.py --lanes-per-env $g --no-cpu-baseline --no-fused --steps 60 --warmup 10 > /dev/null 2> $d.log
  python3 - "$d" "$g" <<'PY'
import csv,glob,sys,collections
d,g=sys.argv[1],sys.argv[2]
f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
if not f: print('no csv', d); sys.exit()
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f[0])):
    if 'step_kernel' in r['Kernel_Name']:
        a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
print('GS',g, {k: round(v[1]/v[0],1) for k,v in acc.items()})
PY
done; done
