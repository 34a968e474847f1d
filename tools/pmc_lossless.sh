# SQ counters of modular_ans_kernel for one 4K lossless stream kind (instructions per sample, where the wavefronts wait)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=${1:-gradient-context-tree}
python3 tools/lossless_one.py $K 1 > gpurun_out/pmcl_prep.log 2>&1   # writes the cached inputs (oracle encoder) outside the profiled runs
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmcl$i -o c -- python3 tools/lossless_one.py $K 2 > gpurun_out/pmcl$i.log 2>&1
  echo "set $i exit=$?"
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(float); n = 0
for f in glob.glob('gpurun_out/pmcl*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'modular_ans_kernel' not in row['Kernel_Name']: continue
        acc[row['Counter_Name']] += float(row['Counter_Value']) / 2    # two decodes per run
samples = 3 * 3840 * 2160
print("stream kind: $K; per decode, per sample (24.9 M samples), and per wavefront (140):")
for k, v in sorted(acc.items()):
    print("  %-28s %14.0f   %8.2f / sample" % (k, v, v / samples))
PY
rm -rf gpurun_out/pmcl1 gpurun_out/pmcl2 gpurun_out/pmcl3 gpurun_out/pmcl4 gpurun_out/pmcl5
