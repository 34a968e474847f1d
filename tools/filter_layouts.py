"""filters + output stage of a synchronous batch of 48 4K frames per output layout, both forms of the streaming kernels
(option no_stream_pairs): python tools/filter_layouts.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
img = synth(3840, 2160, 5)
dec = api.Decoder(0)
n = 48
for layout, src in (("rgba", img), ("rgb", np.ascontiguousarray(img[..., :3])), ("gray", np.ascontiguousarray(img[..., 1:2])), ("graya", np.ascontiguousarray(img[..., [1, 3]]))):
    for dist in (1.0, 2.0):
        data = O.encode(src, distance=dist)
        out = torch.empty((n,) + src.shape, dtype=torch.uint8, device="cuda")
        dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        res = []
        for nop in (0, 1):
            dec.set_option("no_stream_pairs", nop)
            for rep in range(2):
                dec.decode_batch([data] * n, [out[i].data_ptr() for i in range(n)], [dev.data_ptr()] * n)
            res.append(dec.stage_times()["filters+output"])
        dec.set_option("no_stream_pairs", 0)
        print("%-5s distance %.1f: filters + output of %d frames  %.2f ms (two pixels per lane where the layout allows)  %.2f ms (general form)" % (layout, dist, n, res[0], res[1]), flush=True)
