"""BASELINE.json configs[2] rehearsed on ONE GPU: a 16384x16384 RGBA lossy frame (written by the product's own encoder), decoded whole and
then as the 8 group-row bands an 8-GPU run would give one to each rank (decode_frame_band: band + one redundant halo row each side).
Prints per-band times; the bands are checked against the whole-frame decode."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.distributed import decode_frame_band
from pdn_jpegxl_amd.synth import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
base = synth(3840, 2160, 3)
img = np.tile(base, (N // 2160 + 1, N // 3840 + 1, 1))[:N, :N]
bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
t0 = time.perf_counter()
data = api.save_image(bgra, distance=1.0)
print("encode %dx%d: %.2f s, %.1f MB" % (N, N, time.perf_counter() - t0, len(data) / 1e6), flush=True)
del bgra
dec = api.Decoder(0)
whole = torch.empty(N * N * 4, dtype=torch.uint8, device="cuda")
tws = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert dec.decode_batch([data], [whole.data_ptr()]) == [0]
    torch.cuda.synchronize()
    tws.append(time.perf_counter() - t0)
tw = min(tws[1:])
print("whole frame on one GPU: %.1f ms (%.0f MP/s); every call: %s ms; stages of the last: %s"
      % (tw * 1e3, N * N / tw / 1e6, [round(t * 1e3, 1) for t in tws], {k: round(v, 1) for k, v in dec.stage_times().items()}), flush=True)
ref = whole.view(N, N, 4)
world = 8
times = []
for rank in range(world):
    decode_frame_band(dec, data, rank, world)          # warm-up (workspace allocation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    band, (y0, y1) = decode_frame_band(dec, data, rank, world)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    assert bool((band.view(y1 - y0, N, 4) == ref[y0:y1]).all()), rank
print("8 bands, one after another on one GPU: %s ms each; slowest %.1f ms -> %.0f MP/s if the 8 ranks run concurrently (+ one all_gather of %d MiB)"
      % ([round(t * 1e3, 1) for t in times], max(times) * 1e3, N * N / max(times) / 1e6, N * N * 4 // 8 >> 20), flush=True)
