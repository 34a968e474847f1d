"""Helpers shared by the GPU parity tests: run the HIP path through the C-ABI batch entry point."""
import numpy as np


def gpu_decode(dec, files, taps=False, lane_stride=0, resident=False):
    """Returns (list of uint8 HxWxC arrays, decoder) for a batch of .jxl byte strings."""
    import torch
    from pdn_jpegxl_amd import api
    dec.set_option("debug_taps", 1 if taps else 0)
    dec.set_option("lane_stride", lane_stride)
    infos = [api.peek(f) for f in files]
    outs = [torch.empty(i.width * i.height * i.num_channels, dtype=torch.uint8, device="cuda") for i in infos]
    dev_in = None
    keep = []
    if resident:
        for f in files:
            t = torch.zeros(len(f) + 64, dtype=torch.uint8, device="cuda")
            t[:len(f)] = torch.frombuffer(bytearray(f), dtype=torch.uint8).cuda()
            keep.append(t)
        dev_in = [t.data_ptr() for t in keep]
    torch.cuda.synchronize()
    st = dec.decode_batch(files, [o.data_ptr() for o in outs], dev_in)
    assert all(s == 0 for s in st), st
    return [o.cpu().numpy().reshape(i.height, i.width, i.num_channels) for o, i in zip(outs, infos)]


def compare_stages(dec, index, od, tol_lf=1e-5, tol_xyb=2e-4):
    """Compares the HIP stage taps of image `index` with the oracle dump `od` (oracle_lib.Decoded).  Returns a report dict."""
    rep = {}
    w8, h8 = od.w8, od.h8
    ci = dec.read_plane(index, "cellinfo")
    strat = (ci & 0xFF).astype(np.uint8)
    first = ((ci >> 8) & 0x3FF) == 0
    o_s = od.planes["strategy"]
    rep["strategy"] = bool(((o_s & 0x7F) == strat).all() and (((o_s & 0x80) != 0) == first).all())
    rep["raw_quant"] = bool((dec.read_plane(index, "raw_quant").astype(np.int32) == od.planes["raw_quant"]).all())
    rep["sharpness"] = bool((dec.read_plane(index, "sharpness") == od.planes["sharpness"]).all())
    rep["ytox"] = bool((dec.read_plane(index, "ytox") == od.planes["ytox"]).all())
    rep["ytob"] = bool((dec.read_plane(index, "ytob") == od.planes["ytob"]).all())
    for c in range(3):
        rep["lf_quant%d" % c] = bool((dec.read_plane(index, "lf_quant", c) == od.planes["lf_quant"][c]).all())
        rep["lf%d" % c] = float(np.abs(dec.read_plane(index, "lf", c) - od.planes["lf"][c]).max())
        q = dec.read_plane(index, "qcoef", c)
        rep["qcoef%d" % c] = int((q != od.planes["qcoef"][c]).sum())
        wp = w8 * 8
        x = dec.read_plane(index, "xyb_idct", c).reshape(h8 * 8, wp)
        ox = od.planes["xyb_idct"][c]
        h = ox.size // (ox.size // (h8 * 8) if False else 1)
        W = od.pixels.shape[1]
        H = od.pixels.shape[0]
        rep["xyb_idct%d" % c] = float(np.abs(x[:H, :W] - ox.reshape(H, W)).max())
        xf = dec.read_plane(index, "xyb_filtered", c).reshape(h8 * 8, wp)
        rep["xyb_filtered%d" % c] = float(np.abs(xf[:H, :W] - od.planes["xyb_filtered"][c].reshape(H, W)).max())
    if od.planes["alpha"].size:
        rep["alpha"] = bool((dec.read_plane(index, "alpha").astype(np.int32) == od.planes["alpha"]).all())
    return rep
