"""Manual robustness campaign (not a test): random byte corruptions of valid streams must end in Ok or a clean error, never a hang
or a GPU fault.  Run on the GPU box:  timeout -k 10 300 python tools/fuzz_gpu.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import icc_util
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

rng = np.random.default_rng(7)
img = synth(300, 280, 3)
streams = [O.encode(img, distance=1.0), O.encode(img, distance=2.0, strategy_mode=2, seed=4), O.encode(img, lossless=True),
           O.encode(img, lossless=True, lossless_squeeze=True), O.encode(synth(64, 48, 5)),
           O.encode(img.astype(np.uint16) * 257, distance=1.0, bits=16, orientation=6), O.encode(img.astype(np.uint16) * 257, lossless=True, bits=16),
           O.encode((img / 255.0).astype(np.float32), lossless=True, float_samples=32, lossless_predictor=5, lossless_tree=1),
           O.encode(img, distance=1.5, colour=4),
           # round-2 stream features
           O.encode(img, distance=1.0, prefix_codes=True, lz77=True), O.encode(img, distance=1.0, num_passes=3, custom_orders=True),
           O.encode(img, distance=1.0, strategy_mode=2, seed=8, lf_contexts=True, custom_quant_tables=True),
           O.encode(synth(200, 140, 6), distance=1.0, lf_contexts=True, custom_orders=True, prefix_codes=True),
           O.encode(img, lossless=True, prefix_codes=True, lz77=True), O.encode(img[..., :3], lossless=True, cmyk=False, icc=None),
           # round 3: an embedded ICC profile (the ICC stream reader sits in front of everything), premultiplied alpha, AFV labels
           O.encode(img[..., :3], lossless=True, icc=icc_util.matrix_profile("p3", "srgb-para")), O.encode(img, distance=1.0, icc=icc_util.matrix_profile("adobe", "gamma2.2")),
           O.encode(img, lossless=True, premultiplied_alpha=True), O.encode(img, distance=1.0, mislabel_afv=True)]
counts = {}
for si, data in enumerate(streams):
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
        d = bytearray(data)
        lo = 40 + (trial % 7) * 20                 # spare the signature; corrupt headers and payloads alike
        hi = len(d) if trial % 3 else min(len(d), 400)   # a third of the trials stay in the headers (ICC stream, frame header, TOC)
        for _ in range(1 + trial % 4):
            p = int(rng.integers(lo, hi))
            d[p] ^= int(rng.integers(1, 256))
        try:
            api.load_image(bytes(d))
            k = "ok"
        except api.JxlError as e:
            k = e.status
        counts[k] = counts.get(k, 0) + 1
    print("stream", si, counts, flush=True)
print("done", counts)
