"""Times LoadImage on 4K Modular (lossless) streams of three kinds (BASELINE.json configs[4]); a side number, not the headline."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

img = synth(3840, 2160, 2)
rgb = np.ascontiguousarray(img[..., :3])
streams = {
    "product encoder (YCoCg-R, gradient, row-static)": api.save_image(np.ascontiguousarray(img[..., [2, 1, 0, 3]]), lossless=True),
    "oracle, gradient-context tree (generic lane path)": O.encode(rgb, lossless=True, lossless_tree=1, lossless_predictor=5),
    "oracle, weighted predictor + property 15": O.encode(rgb, lossless=True),
    "oracle, Squeeze + weighted predictor": O.encode(rgb, lossless=True, lossless_squeeze=True),
}
for name, data in streams.items():
    api.load_image(data)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        got = api.load_image(data)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    O.decode(data, num_threads=32)
    tc = time.perf_counter() - t0
    print("%-52s %8.1f ms GPU (%.1f MP/s)   CPU oracle 32 threads %8.1f ms   %d bytes" % (name, min(ts) * 1e3, 8.2944 / min(ts), tc * 1e3, len(data)), flush=True)
