"""Decodes frames made of 128x128 / 256x256 / 256x128 DCT varblocks (oracle-encoded) so that rocprofv3 can report the matrix-core
counters of idct_gemm_kernel; checks the pixels against the oracle on the way."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

img = synth(1024, 768, 5)
for s in (21, 24, 25, 18):
    data = O.encode(img, distance=1.0, strategy_mode=3, fixed_strategy=s)
    want = O.decode(data).pixels
    for _ in range(3):
        got = api.load_image(data).pixels
    d = np.abs(got.astype(int) - want.astype(int))
    print("strategy", s, "max diff", d.max(), "mean", d.mean())
    assert d.max() <= 1
