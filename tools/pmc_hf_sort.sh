# hf_decode with and without the size-sorted section order (experiments build): kernel time and vector instructions at batch 96, sync
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in sorted unsorted; do
  if [ $mode = unsorted ]; then export JXLHIP_NO_HF_SORT=1; else unset JXLHIP_NO_HF_SORT; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/hfs_$mode -o c -- python3 bench.py --steps 2 --warmup 1 --batch 96 --no-cpu-baseline --sync-steps > gpurun_out/hfs_$mode.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = 0
for f in glob.glob('gpurun_out/hfs_$mode/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'hf_decode_kernel' not in row['Kernel_Name'] or int(row.get('Grid_Size', 0) or 0) < 64 * 64: continue
        acc[row['Counter_Name']] += float(row['Counter_Value'])
print("$mode", {k: round(v / 1e6, 1) for k, v in acc.items()})
PY
  grep '^{' gpurun_out/hfs_$mode.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  hf_decode ms', d['stage_ms_per_step']['hf_decode'])"
  rm -rf gpurun_out/hfs_$mode
done
