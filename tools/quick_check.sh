# parity subset + default-length bench (through gpurun): the usual loop while tuning a kernel
timeout -k 10 300 python -m pytest tests/test_gpu_formats.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'single', d['single_image']['latency_ms'])"
