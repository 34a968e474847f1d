"""Debug helper: decode the 4K fixtures with both bit-window sizes / several batch sizes and report statuses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdn_jpegxl_amd import api
import json
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
idx = json.load(open(os.path.join(G, "bench_index.json")))
files = [open(os.path.join(G, n + ".jxl"), "rb").read() for n in sorted(idx, key=lambda n: idx[n]["seed"])]
info = api.peek(files[0])
n = info.width * info.height * info.num_channels
outs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(96)]
def run(dec, fs):
    try:
        st = dec.decode_batch(fs, [outs[i].data_ptr() for i in range(len(fs))])
        return "ok"
    except Exception as e:
        return "FAIL " + str(e)[-90:]
mode = sys.argv[1]
if mode == "single":
    for ring in (16, 32):
        os.environ["JXLHIP_HF_RING"] = str(ring)
        dec = api.Decoder(0)
        print("ring", ring, [run(dec, [f]) for f in files], flush=True)
        dec.close()
else:
    os.environ.pop("JXLHIP_HF_RING", None)
    dec = api.Decoder(0)
    for B in (8, 32, 96):
        print("auto ring, batch", B, run(dec, [files[i % 8] for i in range(B)]), flush=True)
