"""Wall time of LoadImage / SaveImage for the small images a host application handles most (BASELINE.json configs[0]: 512x512 RGBA8 lossy
through the C-ABI; plus a one-group 200x150): host buffers, transfers and callbacks included, against the CPU oracle on the same box."""
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


def best(fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for (w, h) in ((512, 512), (200, 150), (1920, 1080)):
    img = synth(w, h, 3)
    bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
    data = O.encode(img, distance=1.0)
    tl = best(lambda: api.load_image(data))
    tc = best(lambda: O.decode(data, num_threads=8), 3)
    ts = best(lambda: api.save_image(bgra, distance=1.0), 5)
    tsl = best(lambda: api.save_image(bgra, lossless=True), 5)
    print("%4dx%-4d LoadImage %6.2f ms (CPU oracle, 8 threads: %6.2f ms)   SaveImage lossy %6.2f ms, lossless %6.2f ms" % (w, h, tl, tc, ts, tsl), flush=True)
