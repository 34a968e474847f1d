for st in 1 2 4; do
  for mode in "--sync-steps" ""; do
    JXLHIP_HF_STRIDE=$st timeout -k 10 300 python bench.py --steps 6 --no-cpu-baseline $mode > gpurun_out/hfstride_$st.log 2>&1 || exit 1
    python - <<PY
import json
l=[x for x in open("gpurun_out/hfstride_$st.log") if x.startswith("{")][-1]
j=json.loads(l); print("stride", $st, "$mode", "ms/step", j["ms_per_step"], {k: round(v,1) for k,v in j["stage_ms_per_step"].items()})
PY
  done
done
