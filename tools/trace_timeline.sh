cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${1:-384}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 4 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/tl.log 2>&1
echo "exit=$?"
F=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
head -2 $F | cut -c1-600
python3 tools/timeline.py $F 1200 > gpurun_out/timeline_b$B.txt
grep '^{' gpurun_out/tl.log | cut -c1-300
rm -rf gpurun_out/tl
