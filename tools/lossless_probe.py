"""Stage times of one 4K Modular (lossless) frame of each kind (device-resident decode)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
img = synth(3840, 2160, 2); rgb = np.ascontiguousarray(img[..., :3])
streams = {"product": api.save_image(np.ascontiguousarray(img[..., [2, 1, 0, 3]]), lossless=True),
           "gradient-context tree": O.encode(rgb, lossless=True, lossless_tree=1, lossless_predictor=5),
           "weighted predictor": O.encode(rgb, lossless=True)}
dec = api.Decoder(0)
for name, data in streams.items():
    info = api.peek(data)
    out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dec.decode_batch([data], [out.data_ptr()]); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("%-24s %.1f ms  %s" % (name, t * 1e3, {k: round(v, 1) for k, v in dec.stage_times().items() if v > 0.05}), flush=True)
