timeout -k 10 200 python bench.py --batch 384 --steps 3 --warmup 1 --no-cpu-baseline --sync-steps > gpurun_out/bench_sync.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/bench_sync.log"):
    if l.startswith("{"):
        d=json.loads(l); print("sync", d["value"], d["ms_per_step"], {k:round(v,2) for k,v in d["stage_ms_per_step"].items()})
PY
