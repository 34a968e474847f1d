"""First-light script (not a pytest): decode one synthetic image on the GPU and print per-stage parity."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
from gpu_helpers import gpu_decode, compare_stages

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (600, 400)
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
img = synth(W, H, 1)
data = O.encode(img, strategy_mode=mode, seed=7)
od = O.decode(data, want_dump=True)
dec = api.Decoder(0)
t = time.time()
out = gpu_decode(dec, [data], taps=True)[0]
print("gpu decode wall", time.time() - t)
rep = compare_stages(dec, 0, od)
print(json.dumps(rep, indent=1))
d = np.abs(out.astype(int) - od.pixels.astype(int))
print("final max diff", d.max(), "count>0", int((d > 0).sum()), "of", d.size)
print(dec.stage_times())
