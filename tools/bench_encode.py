"""Times SaveImage (host buffers in, file bytes out through the Write callback) on the 4K synthetic frame.  Not the headline metric:
a reported side number for BASELINE.json configs[3]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

img = synth(3840, 2160, 2)
bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
for lossless in (False, True):
    api.save_image(bgra, lossless=lossless)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        data = api.save_image(bgra, lossless=lossless)
        ts.append(time.perf_counter() - t0)
    print("lossless=%s: %.1f ms best of 3 (%.1f MP/s), %d bytes (%.2f bpp)" % (lossless, min(ts) * 1e3, 8.2944 / min(ts), len(data), len(data) * 8 / 8.2944e6), flush=True)
