# SQ counter passes (four counters per pass) over a small synchronous bench; folds them into gpurun_out/${R}_pmc_sq_counters_b${B:-8}.json
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r01}
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/sq$i -o c -- python3 bench.py --steps 2 --warmup 1 --batch ${B:-8} --no-cpu-baseline --sync-steps > gpurun_out/sq$i.log 2>&1
  echo "set $i exit=$?"
done
python3 - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/sq*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('jxlhip::', '')
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
keep = {k: dict(v) for k, v in acc.items() if k.endswith('_kernel') or '_kernel<' in k}
json.dump({"source": "rocprofv3 --pmc <4 counters per pass> --kernel-trace, bench.py --steps 2 --warmup 1 --batch 8 --sync-steps; sums over all launches; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles", "kernels": keep}, open('gpurun_out/${R}_pmc_sq_counters_b${B:-8}.json', 'w'), indent=1)
tot_valu = sum(v.get('SQ_INSTS_VALU', 0) for v in keep.values()) or 1
for k, v in keep.items():
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-28s VALU insts %6.1f M (%4.1f%% of all kernels)  MFMA busy %d' % (k[:28], v.get('SQ_INSTS_VALU', 0) / 1e6, 100 * v.get('SQ_INSTS_VALU', 0) / tot_valu, v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)))
    print('%-28s issuing %4.1f%%  parked %4.1f%%  stalled %4.1f%%  ldsconf %4.1f%%' % (k[:28], 100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, v.get('SQ_LDS_IDX_ACTIVE', 1))))
PY
rm -rf gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 gpurun_out/sq4 gpurun_out/sq5
