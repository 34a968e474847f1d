"""BASELINE.json configs[2] at FULL size on one GPU: a 16384x16384 RGBA lossy frame, decoded whole and as the eight group-row bands
an eight-rank run hands out (the RCCL gather itself is covered by the gloo tests).  No oracle at this size: size-independent
properties instead - every band equals its rows of the whole-frame decode bit for bit, alpha is bit-exact against the source, the
colour error against the source is bounded, and repeated tiles of the source decode alike away from tile borders."""
import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

pytestmark = pytest.mark.gpu


def test_16k_frame_whole_and_in_bands():
    import torch
    from pdn_jpegxl_amd.distributed import decode_frame_band
    n = 16384
    base = synth(2048, 2048, 3)
    img = np.tile(base, (n // 2048, n // 2048, 1))
    data = api.save_image(np.ascontiguousarray(img[..., [2, 1, 0, 3]]), distance=1.0)
    info = api.peek(data)
    assert (info.width, info.height, info.num_channels, info.num_groups, info.num_lf_groups) == (n, n, 4, 4096, 64)
    dec = api.Decoder(0)
    whole = torch.empty(n * n * 4, dtype=torch.uint8, device="cuda")
    assert dec.decode_batch([data], [whole.data_ptr()]) == [0]
    ref = whole.view(n, n, 4)
    for rank in range(8):
        band, (y0, y1) = decode_frame_band(dec, data, rank, 8)
        assert (y0, y1) == (rank * 2048, (rank + 1) * 2048)
        assert bool((band.view(y1 - y0, n, 4) == ref[y0:y1]).all()), rank
    out = ref.cpu().numpy()
    dec.close()
    assert np.array_equal(out[..., 3], img[..., 3])                                   # alpha: lossless
    mse = ((out[:4096, :4096, :3].astype(np.float64) - img[:4096, :4096, :3]) ** 2).mean()
    assert 10 * np.log10(255 ** 2 / mse) > 36.0
    # the source repeats every 2048 pixels (= one LF group): interior tiles see identical content and identical neighbours
    assert np.array_equal(out[2048:4096, 2048:4096], out[6144:8192, 10240:12288])
