export JXLHIP_EXTRA_CFLAGS="-DJXLHIP_EXPERIMENTS"
python -c "from pdn_jpegxl_amd import build; build.build()" > gpurun_out/interfere_build.log 2>&1 || { tail -5 gpurun_out/interfere_build.log; exit 1; }
run() {
  env $3 JXLHIP_INTERFERE="$1" python bench.py --steps 4 --sync-steps --no-cpu-baseline 2>gpurun_out/interfere_err.txt > gpurun_out/interfere_tmp.json || { tail -3 gpurun_out/interfere_err.txt; return; }
  python - "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/interfere_tmp.json")); s = d["stage_ms_per_step"]
print("%-70s hf %.1f  alpha_ans %.1f  recon %.1f  filters %.1f" % (sys.argv[1], s["hf_decode"], s["alpha_ans"], s["reconstruct"], s["filters+output"]), flush=True)
PY
}
run "" "hf tables in LDS (75 KB / workgroup), nothing beside" "A=1"
run "" "hf tables from global memory (46 KB / workgroup), nothing beside" "JXLHIP_HF_GLOBAL=1"
run "4,1,56,400" "hf global tables + 56 KB of LDS per CU held" "JXLHIP_HF_GLOBAL=1"
run "4,1,56,400" "hf LDS tables + 56 KB of LDS per CU held" "A=1"
run "" "hf stride 1 (64 lanes / wavefront), nothing beside" "JXLHIP_HF_STRIDE=1"
run "" "hf stride 1, global tables" "JXLHIP_HF_STRIDE=1 JXLHIP_HF_GLOBAL=1"
run "4,1,56,400" "hf stride 1, global tables + 56 KB held" "JXLHIP_HF_STRIDE=1 JXLHIP_HF_GLOBAL=1"
