# timing ablation: which stages bound the pipelined step (JXLHIP_SKIP_STAGES bits: 1 LF, 2 HF, 4 alpha, 8 recon, 16 filters)
for sk in 0 24 7 6 1 16 8 2 4; do
  JXLHIP_SKIP_STAGES=$sk timeout -k 10 300 python bench.py --steps 8 --no-cpu-baseline > gpurun_out/skip_$sk.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/skip_$sk.log") if x.startswith("{")][-1]
j=json.loads(l); print("skip", $sk, "ms/step", j["ms_per_step"], {k: round(v,1) for k,v in j["stage_ms_per_step"].items()})
PY
done
