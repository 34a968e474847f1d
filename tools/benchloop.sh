for B in "$@"; do timeout -k 10 150 python bench.py --batch $B --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cur_b$B.log 2>&1; python - <<PY
import json
ok=False
for l in open("gpurun_out/bench_cur_b$B.log"):
    if l.startswith("{"):
        ok=True
        d=json.loads(l); print("B=$B", d["value"], d["ms_per_step"], {k:round(v,2) for k,v in d["stage_ms_per_step"].items()}, "single", d["single_image"]["latency_ms"], {k:round(v,2) for k,v in d["single_image"]["stage_ms"].items()})
if not ok: print(open("gpurun_out/bench_cur_b$B.log").read()[-1500:])
PY
done
