"""Where a small image's LoadImage time goes: wall time of decode_batch (device output) and of LoadImage next to the GPU stage times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
dec = api.Decoder(0)
for (w, h) in ((200, 150), (512, 512), (1920, 1080)):
    data = O.encode(synth(w, h, 3), distance=1.0)
    out = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); dec.decode_batch([data], [out.data_ptr()]); ts.append((time.perf_counter() - t0) * 1e3)
    st = dec.stage_times()
    tl = []
    for i in range(8):
        t0 = time.perf_counter(); api.load_image(data); tl.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); api.parse_check(data); tp = (time.perf_counter() - t0) * 1e3
    print("%dx%d decode_batch %.2f ms (stages: %s = %.2f)  LoadImage %.2f ms  host parse %.2f ms" % (w, h, min(ts), {k: round(v, 2) for k, v in st.items()}, sum(st.values()), min(tl), tp), flush=True)
