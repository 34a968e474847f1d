# rocprofv3 summaries of the bench command for profiles/ (run on the GPU box through gpurun)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r01}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -o k -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_stats.log 2>&1
echo "stats exit=$?"
cp $(find gpurun_out/${R}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_$C -o c -- python3 bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline --sync-steps > gpurun_out/${R}_pmc_$C.log 2>&1
  echo "pmc $C exit=$?"
  cp $(find gpurun_out/${R}_pmc_$C -name "*counter_collection.csv" | head -1) gpurun_out/${R}_pmc_$C.csv
done
python3 tools/pmc_summary.py gpurun_out/${R}_pmc_FETCH_SIZE.csv gpurun_out/${R}_pmc_WRITE_SIZE.csv 16 gpurun_out/${R}_pmc_traffic.json
head -12 gpurun_out/${R}_kernel_stats.csv | cut -c1-160
grep '^{' gpurun_out/${R}_stats.log | cut -c1-400
