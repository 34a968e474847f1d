# A/B of launch placements in the pipelined 384-frame step (library built with JXLHIP_EXTRA_CFLAGS=-DJXLHIP_EXPERIMENTS; the same value
# must be exported when this runs, api.lib() compares it).  One JSON line per variant into gpurun_out/ab_<name>.json.
run() { name=$1; shift; env "$@" timeout -k 10 240 python bench.py --steps 8 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("%-22s %7.2f ms/step  " % ("$name", d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()})
PY
}
run base X=1 && run hf_global JXLHIP_HF_GLOBAL=1 && run all_global JXLHIP_HF_GLOBAL=1 JXLHIP_ALPHA_GLOBAL=1 JXLHIP_LF_GLOBAL=1 && run pix_on_hf JXLHIP_PIX_ON_HF=1 && run no_overlap JXLHIP_NO_OVERLAP=1 && run hf_lds48 JXLHIP_HF_LDS_KB=48 && run hfglob_pixhf JXLHIP_HF_GLOBAL=1 JXLHIP_PIX_ON_HF=1
