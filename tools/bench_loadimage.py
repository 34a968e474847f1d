"""Wall time of the C-ABI calls a host makes for ONE 4K image (host buffers, PCIe transfers and callbacks included): LoadImage of the
lossy fixture, SaveImage lossy / lossless of the same pixels.  Side numbers for DESIGN.md section 6, not the headline."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

data = open(os.path.join(ROOT, "tests", "golden", "synth_3840x2160_seed2_d1.jxl"), "rb").read()


def best(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


t = best(lambda: api.load_image(data))
print("LoadImage 4K lossy RGBA (%d bytes): %.1f ms = %.1f MP/s (includes the ctypes harness copying 33 MB in setLayerData)" % (len(data), t, 8.2944 / t * 1e3))
bgra = np.ascontiguousarray(synth(3840, 2160, 2)[..., [2, 1, 0, 3]])
t = best(lambda: api.save_image(bgra, distance=1.0), 3)
print("SaveImage 4K lossy d=1.0: %.1f ms" % t)
t = best(lambda: api.save_image(bgra, lossless=True), 3)
print("SaveImage 4K lossless: %.1f ms" % t)
