# Sweep of kPixelChunk (frames that share one set of reconstruction / filter planes): edits the constant, rebuilds and benches on the box.
for c in 32 64 16; do
  sed -i "s/static constexpr int kPixelChunk = [0-9]*;/static constexpr int kPixelChunk = $c;/" pdn_jpegxl_amd/csrc/decoder.cc
  python -c "from pdn_jpegxl_amd import build; build.build()" > /dev/null 2>&1 || { echo build failed; exit 1; }
  for i in 1 2; do python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d[\"stage_ms_per_step\"]; print(\"chunk $c\", d[\"ms_per_step\"], {k:round(v,1) for k,v in s.items() if k in (\"reconstruct\",\"filters+output\",\"hf_decode\")})"; done
done
sed -i "s/static constexpr int kPixelChunk = [0-9]*;/static constexpr int kPixelChunk = 32;/" pdn_jpegxl_amd/csrc/decoder.cc
