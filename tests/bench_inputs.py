"""Input synthesis for bench.py's side workloads (TEST INFRASTRUCTURE: uses the oracle's ENCODER to write streams of the kinds the
product's own encoder does not write - Squeeze, weighted predictor, MA trees over decoded neighbours; BASELINE.json configs[4]).
Nothing here is timed as the product; bench.py only reads the files this writes (cached under a temporary directory)."""
import hashlib
import os
import tempfile

import numpy as np


def cache_dir():
    d = os.path.join(tempfile.gettempdir(), "jxlhip_bench_inputs")
    os.makedirs(d, exist_ok=True)
    return d


def _cached(name, make):
    path = os.path.join(cache_dir(), name)
    if not os.path.exists(path):
        data = make()
        with open(path + ".tmp", "wb") as f:
            f.write(data)
        os.replace(path + ".tmp", path)
    return open(path, "rb").read()


def lossless_4k_streams(threads=32):
    """{label: bytes} of 3840x2160 RGB lossless Modular streams written by the oracle's encoder (seed-2 synthetic picture)."""
    import oracle_lib as O
    from pdn_jpegxl_amd.synth import synth
    rgb = np.ascontiguousarray(synth(3840, 2160, 2)[..., :3])
    tag = hashlib.sha256(rgb.tobytes()).hexdigest()[:12]
    kinds = {
        "squeeze+weighted": dict(lossless_squeeze=True),                               # configs[4] as worded
        "weighted+prop15": dict(),
        "gradient-context-tree": dict(lossless_tree=1, lossless_predictor=5),
    }
    return {k: _cached("lossless4k_%s_%s.jxl" % (k, tag), lambda kw=kw: O.encode(rgb, lossless=True, num_threads=threads, **kw)) for k, kw in kinds.items()}, rgb


def lossy_512():
    """512x512 RGBA8 lossy (distance 1.0) stream of configs[0], written by the oracle's encoder, and its source picture."""
    import oracle_lib as O
    from pdn_jpegxl_amd.synth import synth
    img = synth(512, 512, 1)
    return _cached("lossy512_d1.jxl", lambda: O.encode(img, distance=1.0)), img
