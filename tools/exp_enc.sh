set -e
timeout -k 10 600 python -m pytest tests/test_gpu_encode.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1 || { tail -40 gpurun_out/exp_pytest.log; exit 1; }
tail -2 gpurun_out/exp_pytest.log
timeout -k 10 300 python tools/bench_loadimage.py
