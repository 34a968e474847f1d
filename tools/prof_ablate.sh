cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for A in 0 32 1 2 4 8 3 7; do
  export JXLHIP_ABLATE=$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$A -o r -- python3 bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --sync-steps > gpurun_out/abl$A.log 2>&1
  echo "ablate=$A $(grep -h "recon_tile" $(find gpurun_out/abl$A -name "*kernel_stats.csv") 2>/dev/null | cut -d, -f2-7 | tail -c 120)"
done
