for cfg in "128 1" "128 2" "128 4" "128 16" "192 0" "192 2"; do set -- $cfg; timeout -k 10 200 python bench.py --batch $1 --lane-stride $2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_b$1_s$2.log 2>&1; python - <<PY
import json
ok=False
for l in open("gpurun_out/sweep_b$1_s$2.log"):
    if l.startswith("{"):
        ok=True
        d=json.loads(l); print("B=$1 stride=$2", d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms_per_step"].items()})
if not ok: print("B=$1 stride=$2 FAILED", open("gpurun_out/sweep_b$1_s$2.log").read()[-600:])
PY
done
