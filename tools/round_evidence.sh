# Everything the round's profiles/ directory holds, in one gpurun call (about ten minutes of box time):  bash tools/round_evidence.sh r03
R=${1:-r03}
O=gpurun_out
step() { echo "== $1"; }
step "gpu tests";        timeout -k 10 900 python -m pytest tests -q -m gpu > $O/${R}_pytest_gpu.log 2>&1; tail -2 $O/${R}_pytest_gpu.log
step "bench 4k";         timeout -k 10 400 python bench.py > $O/${R}_bench_b384.json 2> $O/${R}_bench_b384.err; cut -c1-300 $O/${R}_bench_b384.json
step "bench 4k batch 1"; timeout -k 10 300 python bench.py --batch 1 --steps 30 --no-cpu-baseline > $O/${R}_bench_b1.json 2> $O/${R}_bench_b1.err; cut -c1-200 $O/${R}_bench_b1.json
for w in 512-abi 4k-encode 4k-lossless; do step "bench $w"; timeout -k 10 500 python bench.py --workload $w > $O/${R}_bench_$w.json 2> $O/${R}_bench_$w.err; cut -c1-250 $O/${R}_bench_$w.json; done
step "bench 16k-bands";  timeout -k 10 500 python bench.py --workload 16k-bands --steps 5 > $O/${R}_bench_16k_bands.json 2> $O/${R}_bench_16k_bands.err; cut -c1-250 $O/${R}_bench_16k_bands.json
step "bench d=2.0";      timeout -k 10 300 python bench.py --distance 2.0 --steps 6 --no-cpu-baseline > $O/${R}_bench_d2.json 2> $O/${R}_bench_d2.err; cut -c1-200 $O/${R}_bench_d2.json
step "bench d=4.5";      timeout -k 10 300 python bench.py --distance 4.5 --steps 6 --no-cpu-baseline > $O/${R}_bench_d4.5.json 2> $O/${R}_bench_d4.5.err; cut -c1-200 $O/${R}_bench_d4.5.json
step "bench 4k, one batch at a time (every stage alone)"; timeout -k 10 300 python bench.py --steps 10 --sync-steps --no-cpu-baseline > $O/${R}_b384_sync_stages.json 2> $O/${R}_b384_sync_stages.err; cut -c1-200 $O/${R}_b384_sync_stages.json
step "16k one-GPU rehearsal"; timeout -k 10 500 python tools/bench_16k_bands.py > $O/${R}_16k_bands_one_gpu_rehearsal.log 2>&1; tail -3 $O/${R}_16k_bands_one_gpu_rehearsal.log
step "small images";     timeout -k 10 300 python tools/bench_small_images.py > $O/${R}_small_images.log 2>&1; tail -4 $O/${R}_small_images.log
step "fuzz";             timeout -k 10 500 python tools/fuzz_gpu.py 40 > $O/${R}_fuzz_campaign.log 2>&1; tail -2 $O/${R}_fuzz_campaign.log
