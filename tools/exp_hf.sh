# experiment: HF bit-window width x stream priorities at batch 384
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1 || { tail -30 gpurun_out/exp_pytest.log; exit 1; }
tail -2 gpurun_out/exp_pytest.log
for cfg in "32 0" "16 0" "16 1" "16 2" "32 1"; do
  set -- $cfg
  echo "== ring $1 prio $2"
  JXLHIP_HF_RING=$1 JXLHIP_PIX_PRIO=$2 bash tools/benchloop.sh 384
done
