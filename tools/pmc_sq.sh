cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/sq$i -o c -- python3 bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --sync-steps > gpurun_out/sq$i.log 2>&1
  echo "set $i exit=$?"
  python3 - <<PY
import csv, collections
f=[l for l in __import__('glob').glob('gpurun_out/sq$i/**/*counter_collection.csv', recursive=True)]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row['Kernel_Name'].split('(')[0].replace('void ','').replace('jxlhip::','')
    acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
for k in ('recon_tile_kernel<false>','filter_gab_epf1_kernel','hf_decode_kernel<true>','lf_ans_kernel<true>','alpha_ans_kernel<true>'):
    if k in acc: print(k, {c: '%.3g'%v for c,v in acc[k].items()})
PY
done
