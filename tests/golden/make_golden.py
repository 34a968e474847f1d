"""Generates the committed golden fixtures with the CPU oracle (oracle/).

No libjxl, no reference fixtures and no .jxl files exist in the reference tree, so these vectors pin
the oracle against ITSELF over time (regression), not against libjxl: parity with libjxl stays unpinned.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from pdn_jpegxl_amd.synth import synth, synth16  # noqa: E402

CASES = {
    "rgba_64x48_d1": dict(size=(64, 48), seed=11, nch=4, enc=dict(distance=1.0)),
    "rgba_300x280_mix_d2": dict(size=(300, 280), seed=12, nch=4, enc=dict(distance=2.0, strategy_mode=2, seed=5)),
    "rgb_333x257_d1": dict(size=(333, 257), seed=13, nch=3, enc=dict(distance=1.0)),
    "gray_270x300_d3": dict(size=(270, 300), seed=14, nch=1, enc=dict(distance=3.0)),
    "rgba_40x30_lossless": dict(size=(40, 30), seed=15, nch=4, enc=dict(lossless=True)),
    "rgb_300x260_lossless_squeeze": dict(size=(300, 260), seed=16, nch=3, enc=dict(lossless=True, lossless_squeeze=True)),
    "rgba16_96x64_lossless": dict(size=(96, 64), seed=17, nch=4, enc=dict(lossless=True, bits=16)),
    "rgba12_96x64_d1": dict(size=(96, 64), seed=18, nch=4, enc=dict(distance=1.0, bits=12)),
    "rgb_80x60_orient6_d1": dict(size=(80, 60), seed=19, nch=3, enc=dict(distance=1.0, orientation=6)),
}


# The inputs of bench.py (BASELINE.json configs[1]): eight DISTINCT 3840x2160 RGBA8 images, distance 1.0, default heuristics of the
# oracle's encoder (variable transform sizes).  Written by `make_golden.py --bench` into bench_index.json.
BENCH_CASES = {"synth_3840x2160_seed%d_d1" % seed: dict(size=(3840, 2160), seed=seed, nch=4, enc=dict(distance=1.0)) for seed in range(2, 10)}


def source(case):
    w, h = case["size"]
    bits = case["enc"].get("bits", 8)
    img = synth(w, h, case["seed"]) if bits <= 8 else synth16(w, h, case["seed"], bits)
    nch = case["nch"]
    return {4: img, 3: img[..., :3], 1: img[..., 1:2], 2: img[..., [1, 3]]}[nch]


def main():
    bench = "--bench" in sys.argv
    index = {}
    for name, case in (BENCH_CASES if bench else CASES).items():
        src = np.ascontiguousarray(source(case))
        data = O.encode(src, num_threads=1, **case["enc"])
        dec = O.decode(data, num_threads=8 if bench else 1)
        with open(os.path.join(HERE, name + ".jxl"), "wb") as f:
            f.write(data)
        index[name] = dict(size=case["size"], nch=case["nch"], seed=case["seed"], enc=case["enc"], jxl_bytes=len(data),
                           jxl_sha256=hashlib.sha256(data).hexdigest(), pixels_sha256=hashlib.sha256(dec.pixels.tobytes()).hexdigest(),
                           source_sha256=hashlib.sha256(src.tobytes()).hexdigest())
    with open(os.path.join(HERE, "bench_index.json" if bench else "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print(json.dumps(index, indent=1) if not bench else "%d bench fixtures written" % len(index))


if __name__ == "__main__":
    main()
