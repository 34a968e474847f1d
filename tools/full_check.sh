# full GPU test suite + default bench (run through gpurun)
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit=$?"; tail -3 gpurun_out/pytest_gpu_full.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench exit=$?"; grep '^{' gpurun_out/bench_default.log | cut -c1-1500
