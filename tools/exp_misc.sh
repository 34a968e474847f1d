timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_modular.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1; echo "pytest exit=$?"; tail -3 gpurun_out/exp_pytest.log
timeout -k 10 400 python tools/fuzz_gpu.py 40 > gpurun_out/fuzz.log 2>&1; echo "fuzz exit=$?"; tail -3 gpurun_out/fuzz.log
bash tools/benchloop.sh 384 1
