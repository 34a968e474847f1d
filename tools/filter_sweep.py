"""Randomised cross-check of the two forms of the streaming filter kernels (not a test): random sizes, distances, layouts and
Gaborish on / off; the pair kernels against the general form (option no_stream_pairs) and against the oracle.
  timeout -k 10 600 python tools/filter_sweep.py [cases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
rng = np.random.default_rng(11)
dec = api.Decoder(0)
def decode(data, shape):
    out = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dec.decode_batch([data], [out.data_ptr()], [dev.data_ptr()])
    return out.cpu().numpy()
worst = 0.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for case in range(n):
    w = int(rng.integers(4, 360)) * 2
    h = int(rng.integers(1, 420))
    dist, iters = [(1.0, 1), (2.0, 2), (4.5, 3)][case % 3]
    gab = bool(case % 5)
    rgb = bool((case // 3) % 2)
    img = synth(w, h, 100 + case)
    src = np.ascontiguousarray(img[..., :3]) if rgb else img
    data = O.encode(src, distance=dist, epf_iters=iters, gaborish=gab)
    ref = O.decode(data).pixels
    a = decode(data, src.shape)
    dec.set_option("no_stream_pairs", 1)
    b = decode(data, src.shape)
    dec.set_option("no_stream_pairs", 0)
    d_ab = np.abs(a.astype(int) - b.astype(int)); d_ar = np.abs(a.astype(int) - ref.astype(int))
    ok = d_ab.max() <= 1 and (d_ab > 0).mean() < 2e-3 and d_ar.max() <= 1 and (d_ar > 0).mean() <= 2e-3
    worst = max(worst, float((d_ar > 0).mean()))
    print("%3d  %4dx%-4d d=%.1f iters %d gab %d %s  pairs vs general: max %d frac %.5f   vs oracle: max %d frac %.5f  %s" % (
        case, w, h, dist, iters, gab, "rgb " if rgb else "rgba", d_ab.max(), (d_ab > 0).mean(), d_ar.max(), (d_ar > 0).mean(), "ok" if ok else "MISMATCH"), flush=True)
    if not ok: sys.exit(1)
print("all", n, "cases ok; worst fraction of samples off by one against the oracle: %.5f" % worst)
