# Which resource is each stage sensitive to?  One batch at a time (every stage alone on the chip) with a probe kernel beside it that
# occupies ONE resource from the start of the HF stage on (experiments build; kernels.hip interfere_kernel).  Prints the stage times.
export JXLHIP_EXTRA_CFLAGS="-DJXLHIP_EXPERIMENTS"
python -c "from pdn_jpegxl_amd import build; build.build()" > gpurun_out/interfere_build.log 2>&1 || { tail -5 gpurun_out/interfere_build.log; exit 1; }
run() {
  JXLHIP_INTERFERE="$1" python bench.py --steps 4 --sync-steps --no-cpu-baseline 2>gpurun_out/interfere_err.txt > gpurun_out/interfere_tmp.json || { tail -3 gpurun_out/interfere_err.txt; return; }
  python - "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/interfere_tmp.json")); s = d["stage_ms_per_step"]
print("%-58s hf %.1f  alpha_ans %.1f  alpha_finish %.1f  recon %.1f  filters %.1f" % (sys.argv[1], s["hf_decode"], s["alpha_ans"], s["alpha_finish"], s["reconstruct"], s["filters+output"]), flush=True)
PY
}
run "" "nothing beside the stages"
run "1,1,1,400" "dependent vector chain, 1 wavefront / SIMD"
run "1,4,1,400" "dependent vector chains, 4 wavefronts / SIMD"
run "2,1,1,400" "independent vector instructions, 1 wavefront / SIMD"
run "2,4,1,400" "independent vector instructions, 4 wavefronts / SIMD"
run "3,1,1,400" "LDS traffic, 1 wavefront / SIMD"
run "3,4,1,400" "LDS traffic, 4 wavefronts / SIMD"
run "4,1,50,400" "LDS capacity: 50 KB per CU held"
run "4,2,50,400" "LDS capacity: 100 KB per CU held"
run "5,1,1,400" "registers: 256 per SIMD held (of 512)"
run "6,2,1,400" "L2 / HBM traffic: streaming copy, 2 wavefronts / SIMD"
