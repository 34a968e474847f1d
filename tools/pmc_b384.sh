# SQ counters of the entropy kernels at the bench batch size (384), synchronous steps; prints per-kernel ratios
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/sqb$i -o c -- python3 bench.py --steps 1 --warmup 0 --batch 384 --no-cpu-baseline --sync-steps > gpurun_out/sqb$i.log 2>&1
  echo "set $i exit=$?"
done
python3 - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/sqb*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('jxlhip::', '')
        if int(row.get('Grid_Size', 0) or 0) < 64 * 256: continue   # the single-image launches
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
keep = {k: dict(v) for k, v in acc.items() if k.endswith('_kernel') or '_kernel<' in k}
json.dump({"source": "rocprofv3 --pmc, bench.py --steps 1 --warmup 0 --batch 384 --sync-steps, launches of the 384-image batches only", "kernels": keep}, open('gpurun_out/${R:-r03}_pmc_sq_counters_b384.json', 'w'), indent=1)
for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-28s VALU insts %8.1f M  valu-active/busy %5.2f  wave-cycles/busy %5.2f  issuing %4.1f%%  ldsconf %4.1f%%  lds insts %7.1f M' % (k[:28], v.get('SQ_INSTS_VALU', 0) / 1e6, v.get('SQ_ACTIVE_INST_VALU', 0) * 4 / max(1, v.get('SQ_BUSY_CYCLES', 1)), wc * 4 / max(1, v.get('SQ_BUSY_CYCLES', 1)), 100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, v.get('SQ_LDS_IDX_ACTIVE', 1)), v.get('SQ_INSTS_LDS', 0) / 1e6))
PY
rm -rf gpurun_out/sqb1 gpurun_out/sqb2 gpurun_out/sqb3
