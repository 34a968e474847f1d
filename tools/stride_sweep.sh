for cfg in "$@"; do set -- ${cfg/:/ }; B=$1; S=$2; timeout -k 10 200 python bench.py --batch $B --lane-stride $S --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_b${B}_s$S.log 2>&1; python - <<PY
import json
ok=False
for l in open("gpurun_out/sweep_b${B}_s$S.log"):
    if l.startswith("{"):
        ok=True
        d=json.loads(l); print("B=$B stride=$S", d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms_per_step"].items()})
if not ok: print("B=$B stride=$S FAILED", open("gpurun_out/sweep_b${B}_s$S.log").read()[-600:])
PY
done
