set -e
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_modular.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1 || { tail -30 gpurun_out/exp_pytest.log; exit 1; }
tail -2 gpurun_out/exp_pytest.log
echo "== default"; bash tools/benchloop.sh 384
