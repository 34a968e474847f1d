cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for A in 0 1 2 4 6 7; do
  export JXLHIP_ABLATE_F=$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablf$A -o r -- python3 bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --sync-steps > gpurun_out/ablf$A.log 2>&1
  echo "ablate_f=$A $(grep -h "filter_gab_epf1" $(find gpurun_out/ablf$A -name "*kernel_stats.csv") 2>/dev/null | cut -d, -f2-7 | tail -c 100)"
done
