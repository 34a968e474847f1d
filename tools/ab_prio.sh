# A/B of the wavefront issue priorities (s_setprio) of the serial decoders and of the pixel kernels in the pipelined step.
# Rebuilds on the box per variant: bash tools/ab_prio.sh "3 0" "0 3" "0 0" ...
for V in "$@"; do
  set -- $V
  export JXLHIP_EXTRA_CFLAGS="-DJXLHIP_PRIO_SERIAL=$1 -DJXLHIP_PRIO_PIXEL=$2"
  python -c "from pdn_jpegxl_amd import build; build.build()" > gpurun_out/ab_prio_build.log 2>&1 || { echo build failed; exit 1; }
  for rep in 1 2; do
    python bench.py --steps 30 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_prio_tmp.json || exit 1
    python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_prio_tmp.json")); s = d["stage_ms_per_step"]
print("serial %s pixel %s: %.2f ms/step | lf %.1f+%.1f hf %.1f alpha %.1f+%.1f recon %.1f filters %.1f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], s["lf_ans"], s["lf_finish+pixels"], s["hf_decode"], s["alpha_ans"], s["alpha_finish"], s["reconstruct"], s["filters+output"]), flush=True)
PY
  done
done
