set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1 || { tail -30 gpurun_out/exp_pytest.log; exit 1; }
tail -2 gpurun_out/exp_pytest.log
echo "== default"; bash tools/benchloop.sh 384
for st in 2 4; do
  echo "== alpha stride $st"
  JXLHIP_ALPHA_STRIDE=$st bash tools/benchloop.sh 384
done
