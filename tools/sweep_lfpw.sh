for v in 1 2 4; do
  JXLHIP_LF_PER_WAVE=$v timeout -k 10 300 python bench.py --steps 6 --no-cpu-baseline > gpurun_out/lfpw_$v.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/lfpw_$v.log") if x.startswith("{")][-1]
j=json.loads(l); print("lf_per_wave", $v, "ms/step", j["ms_per_step"], {k: round(v,1) for k,v in j["stage_ms_per_step"].items()}, "single", j["single_image"]["latency_ms"], round(j["single_image"]["stage_ms"]["lf_ans"],1))
PY
done
