# Spatial partition experiment: entropy streams confined to N CUs (and optionally the pixel stream to the other M); experiments build.
export JXLHIP_EXTRA_CFLAGS="-DJXLHIP_EXPERIMENTS"
python -c "from pdn_jpegxl_amd import build; build.build()" > gpurun_out/ab_cumask_build.log 2>&1 || { echo build failed; exit 1; }
for V in "$@"; do
  set -- $V
  for rep in 1 2; do
    JXLHIP_ENTROPY_CUS=$1 JXLHIP_PIXEL_CUS=$2 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_cumask_tmp.json || exit 1
    python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_cumask_tmp.json")); s = d["stage_ms_per_step"]
print("entropy CUs %s pixel CUs %s: %.2f ms/step | lf %.1f+%.1f hf %.1f alpha %.1f+%.1f recon %.1f filters %.1f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], s["lf_ans"], s["lf_finish+pixels"], s["hf_decode"], s["alpha_ans"], s["alpha_finish"], s["reconstruct"], s["filters+output"]), flush=True)
PY
  done
done
