for kb in 80 100 128 150; do
  JXLHIP_HF_LDS_KB=$kb timeout -k 10 300 python bench.py --steps 8 --no-cpu-baseline > gpurun_out/sweep_hf_$kb.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/sweep_hf_$kb.log") if x.startswith("{")][-1]
j=json.loads(l); print($kb, j["ms_per_step"], {k: round(v,1) for k,v in j["stage_ms_per_step"].items()})
PY
done
