# Generic A/B of experiment knobs in the pipelined step (experiments build): bash tools/ab_env.sh "" "JXLHIP_HF_LDS_KB=96" "A=1 B=2" ...
# (bench flags through BENCH_ARGS)
export JXLHIP_EXTRA_CFLAGS="-DJXLHIP_EXPERIMENTS"
python -c "from pdn_jpegxl_amd import build; build.build()" > gpurun_out/ab_env_build.log 2>&1 || { echo build failed; exit 1; }
for V in "$@"; do
  for rep in 1 2; do
    env $V python bench.py --steps ${STEPS:-30} --no-cpu-baseline $BENCH_ARGS 2>gpurun_out/ab_env_err.txt > gpurun_out/ab_env_tmp.json || { tail -3 gpurun_out/ab_env_err.txt; exit 1; }
    python - "$V" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_env_tmp.json")); s = d["stage_ms_per_step"]
print("%-40s %.2f ms/step | lf %.1f+%.1f hf %.1f alpha %.1f+%.1f recon %.1f filters %.1f" % (sys.argv[1] or "(default)", d["ms_per_step"], s["lf_ans"], s["lf_finish+pixels"], s["hf_decode"], s["alpha_ans"], s["alpha_finish"], s["reconstruct"], s["filters+output"]), flush=True)
PY
  done
done
