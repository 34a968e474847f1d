"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): integer/byte work bit-exact — quantised LF, block metadata, quantised HF coefficients,
alpha; float stages within the tolerances written below; final 8-bit samples within 1 LSB per channel.
"""
import os

import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth
from gpu_helpers import compare_stages, gpu_decode

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_LF = 2e-6          # dequantised LF samples (|values| <= ~1)
TOL_XYB = 5e-5         # XYB samples after IDCT / loop filters: float32 summation-order differences only
MAX_LSB = 1            # final u8 samples
MAX_FRAC_DIFF = 2e-3   # fraction of u8 samples allowed to differ by that one LSB (rounding boundaries)


def check_report(rep):
    for k, v in rep.items():
        if isinstance(v, bool):
            assert v, k
        elif k.startswith("qcoef"):
            assert v == 0, (k, v)
        elif k.startswith("lf"):
            assert v <= TOL_LF, (k, v)
        else:
            assert v <= TOL_XYB, (k, v)


def check_pixels(out, ref):
    assert out.shape == ref.shape
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= MAX_LSB, int(d.max())
    assert (d > 0).mean() <= MAX_FRAC_DIFF, float((d > 0).mean())
    if out.shape[2] in (2, 4):
        assert (out[..., -1] == ref[..., -1]).all()  # alpha: bit-exact


def run_case(dec, oracle, img, **enc):
    data = oracle.encode(img, **enc)
    od = oracle.decode(data, want_dump=True)
    out = gpu_decode(dec, [data], taps=True)[0]
    check_report(compare_stages(dec, 0, od))
    check_pixels(out, od.pixels)
    return data, od


def test_default_heuristics_rgba(gpu_decoder, oracle):
    run_case(gpu_decoder, oracle, synth(600, 400, 1))


def test_random_mix_of_every_transform(gpu_decoder, oracle):
    data, od = run_case(gpu_decoder, oracle, synth(776, 520, 2), strategy_mode=2, seed=7)
    used = set((od.planes["strategy"][od.planes["strategy"] >= 0x80] & 0x7F).tolist())
    assert len(used) >= 18, used


@pytest.mark.parametrize("s", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 19, 20, 21, 22, 23, 24, 25, 26])
def test_each_transform(gpu_decoder, oracle, s):
    run_case(gpu_decoder, oracle, synth(520, 300, 3), strategy_mode=3, fixed_strategy=s)


@pytest.mark.parametrize("size", [(257, 300), (1024, 264), (264, 1100), (300, 257), (2100, 300)])
def test_ragged_sizes(gpu_decoder, oracle, size):
    run_case(gpu_decoder, oracle, synth(size[0], size[1], 4))


@pytest.mark.parametrize("nch", [1, 2, 3])
def test_channel_layouts(gpu_decoder, oracle, nch):
    img = synth(400, 300, 5)
    run_case(gpu_decoder, oracle, np.ascontiguousarray({3: img[..., :3], 1: img[..., 1:2], 2: img[..., [1, 3]]}[nch]))


@pytest.mark.parametrize("epf", [0, 1, 2, 3])
@pytest.mark.parametrize("gab", [True, False])
def test_loop_filter_settings(gpu_decoder, oracle, epf, gab):
    run_case(gpu_decoder, oracle, synth(400, 300, 6), distance=2.5, epf_iters=epf, gaborish=gab)


@pytest.mark.parametrize("dist", [0.3, 1.0, 4.0, 12.0])
def test_distances(gpu_decoder, oracle, dist):
    run_case(gpu_decoder, oracle, synth(400, 300, 7), distance=dist)


def test_without_adaptive_lf_smoothing(gpu_decoder, oracle):
    run_case(gpu_decoder, oracle, synth(400, 300, 8), adaptive_lf_smoothing=False)


def test_noise_image_many_tokens(gpu_decoder, oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (300, 420, 4), dtype=np.uint8)
    run_case(gpu_decoder, oracle, img, distance=0.5)


def test_flat_image_zero_tokens(gpu_decoder, oracle):
    img = np.full((300, 420, 4), 200, np.uint8)
    img[..., 3] = 255
    img[0, 0, 3] = 254
    run_case(gpu_decoder, oracle, img)


def test_batch_and_lane_mappings_agree(gpu_decoder, oracle):
    files, refs = [], []
    for i, (w, h) in enumerate([(600, 400), (300, 700), (1030, 520)]):
        data = oracle.encode(synth(w, h, 20 + i), strategy_mode=2 if i == 1 else 0, seed=i)
        files.append(data)
        refs.append(oracle.decode(data).pixels)
    base = None
    for stride in (64, 8, 1):
        outs = gpu_decode(gpu_decoder, files, lane_stride=stride)
        for o, r in zip(outs, refs):
            check_pixels(o, r)
        if base is None:
            base = outs
        else:
            for a, b in zip(base, outs):
                assert (a == b).all()  # the mapping of sections to lanes must not change a single bit


def test_device_resident_input(gpu_decoder, oracle):
    data = oracle.encode(synth(500, 400, 9))
    a = gpu_decode(gpu_decoder, [data])[0]
    b = gpu_decode(gpu_decoder, [data], resident=True)[0]
    assert (a == b).all()


def test_golden_fixtures_on_gpu(gpu_decoder, oracle):
    for name in ("rgba_300x280_mix_d2", "rgb_333x257_d1", "gray_270x300_d3"):
        data = open(os.path.join(GOLD, name + ".jxl"), "rb").read()
        check_pixels(gpu_decode(gpu_decoder, [data])[0], oracle.decode(data).pixels)


def test_full_size_4k_fixture(gpu_decoder, oracle):
    """BASELINE.json configs[1]: 3840x2160 RGBA8 lossy d=1.  Pixel parity against the oracle plus the
    size-independent properties: alpha bit-exact and bounded error against the SOURCE image."""
    data = open(os.path.join(GOLD, "synth_3840x2160_seed2_d1.jxl"), "rb").read()
    out = gpu_decode(gpu_decoder, [data])[0]
    ref = oracle.decode(data, num_threads=8).pixels
    check_pixels(out, ref)
    src = synth(3840, 2160, 2)
    assert (out[..., 3] == src[..., 3]).all()
    mse = ((out[..., :3].astype(np.float64) - src[..., :3]) ** 2).mean()
    assert 10 * np.log10(255 ** 2 / mse) > 38.0


def test_corrupt_stream_is_reported_not_crashing(gpu_decoder, oracle):
    data = bytearray(oracle.encode(synth(600, 400, 1)))
    info = api.peek(bytes(data))
    import torch
    out = torch.empty(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
    # flip bytes in the middle of the last pass group: headers and TOC stay valid
    for k in range(len(data) - 3000, len(data) - 2900):
        data[k] ^= 0x5A
    with pytest.raises(api.FormatError) as e:
        gpu_decoder.decode_batch([bytes(data)], [out.data_ptr()])
    assert e.value.status == "DecodeError"


# ---------------------------------------------------------------- band-sharded decode of one frame (multi-GPU layout)
@pytest.mark.parametrize("distance,world", [(1.0, 2), (1.0, 3), (4.5, 2), (4.5, 8)])
def test_band_decode_equals_whole_frame(oracle, gpu_decoder, distance, world):
    """Every rank's band (decoded with one redundant halo group row each side) must be bit-identical to the same rows of the
    whole-frame decode: Gaborish + up to three EPF iterations read across the band boundary.  The ranks are emulated one after
    another on the one GPU of the test box; the gather itself is covered by the gloo test in test_distributed.py."""
    import torch
    from pdn_jpegxl_amd.distributed import decode_frame_band
    img = synth(520, 1400, 31)          # 3 x 6 groups
    data = oracle.encode(img, distance=distance)
    whole = torch.empty(img.size, dtype=torch.uint8, device="cuda")
    assert gpu_decoder.decode_batch([data], [whole.data_ptr()]) == [0]
    whole = whole.cpu().numpy().reshape(img.shape)
    rows = 0
    for rank in range(world):
        band, (y0, y1) = decode_frame_band(gpu_decoder, data, rank, world)
        got = band.cpu().numpy().reshape(y1 - y0, img.shape[1], 4)
        assert (got == whole[y0:y1]).all(), (rank, y0, y1)
        rows += y1 - y0
    assert rows == img.shape[0]


# ---------------------------------------------------------------- frames that fit one group (one TOC entry, one bit stream)
@pytest.mark.parametrize("size,kw", [((64, 48), {}), ((256, 256), {}), ((200, 120), dict(distance=3.0)), ((33, 250), dict(strategy_mode=2, seed=5)),
                                     ((8, 8), {}), ((1, 1), {})])
@pytest.mark.parametrize("layout", ["rgba", "rgb", "gray"])
def test_single_group_frames(oracle, size, kw, layout):
    w, h = size
    img = synth(w, h, 41)
    src = np.ascontiguousarray({"rgba": img, "rgb": img[..., :3], "gray": img[..., 1:2]}[layout])
    data = oracle.encode(src, **kw)
    ref = oracle.decode(data).pixels
    got = api.load_image(data)
    assert got.pixels.shape == ref.shape
    d = np.abs(got.pixels.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 0.01
    if layout == "rgba":
        assert (got.pixels[..., 3] == img[..., 3]).all()


# ---------------------------------------------------------------- entropy-coding variants and explicit quantisation tables (row a7)
@pytest.mark.parametrize("opts", [dict(prefix_codes=True), dict(lz77=True), dict(prefix_codes=True, lz77=True)], ids=["prefix", "lz77", "prefix+lz77"])
def test_prefix_coded_and_lz77_streams(gpu_decoder, oracle, opts):
    """Every section stream (LF, HF metadata, HF coefficients, alpha) written with prefix codes instead of ANS and / or LZ77 (runs and
    copies from one row up, so both the plain and the special two-dimensional distances occur): same stage-by-stage parity bar."""
    img = synth(640, 520, 71)
    img[40:200, 50:500, 3] = 255             # long runs in the alpha stream
    img[300:360, :, :] = img[300:301, :, :]  # repeated rows: copies at distance = channel width
    run_case(gpu_decoder, oracle, img, **opts)


@pytest.mark.parametrize("opts", [dict(prefix_codes=True), dict(lz77=True), dict(prefix_codes=True, lz77=True)], ids=["prefix", "lz77", "prefix+lz77"])
@pytest.mark.parametrize("squeeze", [False, True])
def test_prefix_coded_and_lz77_lossless(gpu_decoder, oracle, opts, squeeze):
    img = synth(600, 540, 72)
    img[100:180, 60:400, :] = img[100:101, 60:61, :]   # a flat rectangle
    img[300:340, :, :] = img[300:301, :, :]
    data = oracle.encode(img, lossless=True, lossless_squeeze=squeeze, **opts)
    out = gpu_decode(gpu_decoder, [data])[0]
    assert (out == img).all()


def test_explicit_quantisation_tables(gpu_decoder, oracle):
    """All seventeen dequantisation tables signalled explicitly (identity / DCT2 / DCT4 / DCT4x8 / distance-band encodings with scaled
    parameters), every transform in use: the weights come from the stream, not from the library defaults."""
    data, od = run_case(gpu_decoder, oracle, synth(776, 520, 73), strategy_mode=2, seed=11, custom_quant_tables=True)
    plain = oracle.decode(oracle.encode(synth(776, 520, 73), strategy_mode=2, seed=11)).pixels
    assert (plain != od.pixels).mean() > 0.05      # the tables really differ from the defaults


def test_corrupt_prefix_and_lz77_streams_fail_cleanly(gpu_decoder, oracle):
    """Bit flips in the section data: the decode either ends in an error status, or - when the flips only touched raw bits /
    padding, which no entropy coder can notice - it yields what the oracle yields for the same damaged stream.  Nothing else counts
    as a pass (in particular no exception of the harness), and nothing may crash or hang."""
    import torch
    img = synth(520, 300, 74)
    data = bytearray(oracle.encode(img, prefix_codes=True, lz77=True))
    rng = np.random.default_rng(3)
    detected = 0
    for k in range(12):
        d = bytearray(data)
        for p in rng.integers(len(d) // 3, len(d), 6):
            d[p] ^= 1 << int(rng.integers(0, 8))
        d = bytes(d)
        info = api.peek(d)
        out = torch.zeros(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
        st = gpu_decoder.decode_batch([d], [out.data_ptr()], None, raise_on_error=False)
        if st[0] != 0:
            assert st[0] == api.DECODER_STATUS.index("DecodeError"), st
            detected += 1
            continue
        try:
            ref = oracle.decode(d).pixels
        except oracle.OracleError as e:
            pytest.fail("stream %d: status Ok for a stream the oracle refuses (%s)" % (k, e))
        check_pixels(out.cpu().numpy().reshape(ref.shape), ref)
    assert detected >= 1


def test_afv_varblocks_are_refused_loudly(oracle):
    """AFV0..AFV3 (strategy ids 14..17) are not built: a stream that labels blocks so must end in DecodeError with a message that
    says so - through LoadImage, the call the reference's host makes (Decoder/JxlDecoder.cpp:252, errors :837-849) - never in
    status Ok with unwritten blocks.  The stream comes from the oracle's writer with some 8x8 blocks labelled AFV (refusal test
    switch; the oracle's own decoder refuses AFV as well)."""
    for size in ((520, 300), (200, 150)):          # several groups / a one-group frame
        data = oracle.encode(synth(size[0], size[1], 3), mislabel_afv=True)
        with pytest.raises(oracle.OracleError):
            oracle.decode(data)
        with pytest.raises(api.FormatError) as e:
            api.load_image(data)
        assert e.value.status == "DecodeError"
        assert "AFV" in str(e.value)
    # the same picture without the labels decodes
    api.load_image(oracle.encode(synth(200, 150, 3)))


def test_band_options_are_validated(gpu_decoder, oracle):
    """A negative first band row would put pixel rows in front of the caller's band buffer (advisor, round 2): the option is refused,
    and a band that starts past the frame fails the frame."""
    import torch
    assert gpu_decoder.set_option("band_first_row", -1) == 0
    assert gpu_decoder.set_option("band_rows", -5) == 0
    assert gpu_decoder.set_option("lane_stride", 3) == 0 and gpu_decoder.set_option("lane_stride", 128) == 0
    data = oracle.encode(synth(300, 520, 5))
    info = api.peek(data)
    out = torch.zeros(info.width * info.height * info.num_channels, dtype=torch.uint8, device="cuda")
    try:
        assert gpu_decoder.set_option("band_first_row", 7) and gpu_decoder.set_option("band_rows", 1)
        st = gpu_decoder.decode_batch([data], [out.data_ptr()], None, raise_on_error=False)
        assert st[0] == api.DECODER_STATUS.index("DecodeError")
    finally:
        gpu_decoder.set_option("band_rows", 0)
        gpu_decoder.set_option("band_first_row", 0)
    check_pixels(gpu_decode(gpu_decoder, [data])[0], oracle.decode(data).pixels)


@pytest.mark.parametrize("passes", [2, 3])
@pytest.mark.parametrize("extra", [dict(), dict(strategy_mode=2, seed=5), dict(prefix_codes=True, lz77=True)], ids=["plain", "varblocks", "prefix+lz77"])
def test_progressive_passes(gpu_decoder, oracle, passes, extra):
    """Progressive frames: the quantised coefficients arrive in 2 / 3 passes (own code and coefficient orders each, shifted shares);
    the sum of the passes is what a one-pass frame carries, so the pixels equal the one-pass decode and every stage tap the oracle's."""
    img = synth(700, 530, 75 + passes)
    data, od = run_case(gpu_decoder, oracle, img, num_passes=passes, **extra)
    one = oracle.decode(oracle.encode(img, **extra)).pixels
    assert (od.pixels == one).all()
    check_pixels(gpu_decode(gpu_decoder, [data])[0], one)


def test_progressive_single_group_and_batch(gpu_decoder, oracle):
    """A progressive frame that fits one group still has a table of contents (one section per pass); progressive and plain frames
    share a batch."""
    small = synth(200, 120, 81)
    big = synth(600, 300, 82)
    files = [oracle.encode(small, num_passes=2), oracle.encode(big), oracle.encode(big, num_passes=3), oracle.encode(small)]
    outs = gpu_decode(gpu_decoder, files)
    for f, o in zip(files, outs):
        check_pixels(o, oracle.decode(f).pixels)


@pytest.mark.parametrize("extra", [dict(), dict(strategy_mode=2, seed=9), dict(num_passes=2), dict(prefix_codes=True, lz77=True), dict(custom_quant_tables=True)],
                         ids=["plain", "varblocks", "two-passes", "prefix+lz77", "own-tables"])
def test_custom_coefficient_orders(gpu_decoder, oracle, extra):
    """Coefficient orders written in the stream (per order bucket and channel: positions sorted by how often they are non-zero, as an
    encoder that adapts its scan to the image does) instead of the natural zig-zags: the scan lists are built per image."""
    img = synth(640, 530, 91)
    data, od = run_case(gpu_decoder, oracle, img, custom_orders=True, **extra)
    if "custom_quant_tables" not in extra:
        check_pixels(gpu_decode(gpu_decoder, [data])[0], oracle.decode(oracle.encode(img, **extra)).pixels)


def test_custom_coefficient_orders_in_a_one_group_frame(gpu_decoder, oracle):
    """HfGlobal of a one-section frame is parsed after the LF stage has run on the GPU (its start is only known then); the orders
    it carries still reach the scan lists."""
    img = synth(220, 140, 92)
    run_case(gpu_decoder, oracle, img, custom_orders=True)
    run_case(gpu_decoder, oracle, img, custom_orders=True, strategy_mode=2, seed=4)


@pytest.mark.parametrize("opts", [dict(prefix_codes=True), dict(lz77=True), dict(prefix_codes=True, lz77=True)], ids=["prefix", "lz77", "prefix+lz77"])
def test_prefix_and_lz77_in_a_one_group_frame(gpu_decoder, oracle, opts):
    """One-section frames: the LF stage that finds where HfGlobal starts reads the same general symbol streams."""
    img = synth(230, 150, 93)
    img[20:90, 30:200, 3] = 255
    img[100:130, :, :] = img[100:101, :, :]
    run_case(gpu_decoder, oracle, img, **opts)
    out = gpu_decode(gpu_decoder, [oracle.encode(img, lossless=True, **opts)])[0]
    assert (out == img).all()


@pytest.mark.parametrize("extra", [dict(), dict(strategy_mode=2, seed=6), dict(num_passes=2, custom_orders=True)], ids=["plain", "varblocks", "passes+orders"])
def test_block_contexts_from_lf_and_quant_field_thresholds(gpu_decoder, oracle, extra):
    """A block-context map that depends on the quantised LF of the block (thresholds per channel) and on the quant field, as adaptive
    encoders write it, instead of the default map.  [How the three LF indices combine is recalled from the format, not pinned.]"""
    img = synth(700, 520, 95)
    data, od = run_case(gpu_decoder, oracle, img, lf_contexts=True, **extra)
    check_pixels(gpu_decode(gpu_decoder, [data])[0], oracle.decode(oracle.encode(img, **extra)).pixels)
    small = synth(210, 150, 96)
    run_case(gpu_decoder, oracle, small, lf_contexts=True)


def test_scalar_unit_loops_match_the_vector_loops(gpu_decoder, oracle):
    """One-section wavefronts decode on the scalar unit (per-residue tables through the scalar cache, counted HF token loop); many
    sections per wavefront use the vector loops.  Same streams, both shapes of launch: the outputs must be byte-identical.
    Covers a lossy RGBA frame (LF, HF, alpha rows), a product-encoded lossless frame and an oracle lossless frame with West rows."""
    img = synth(700, 520, 11)
    streams = [oracle.encode(img, distance=1.0),
               api.save_image(np.ascontiguousarray(img[..., [2, 1, 0, 3]]), lossless=True),
               oracle.encode(np.ascontiguousarray(img[..., :3]), lossless=True, lossless_tree=0, lossless_predictor=1)]
    scalar = [gpu_decode(gpu_decoder, [s])[0] for s in streams]
    try:
        assert gpu_decoder.set_option("no_direct", 1)      # no per-residue tables: the vector row loops on one lane
        vector1 = [gpu_decode(gpu_decoder, [s])[0] for s in streams]
        assert gpu_decoder.set_option("mod_lanes64", 1)    # Modular sections 64 to a wavefront
        vector2 = [gpu_decode(gpu_decoder, [s], lane_stride=2)[0] for s in streams]   # HF / alpha sections 32 to a wavefront
    finally:
        gpu_decoder.set_option("no_direct", 0)
        gpu_decoder.set_option("mod_lanes64", 0)
    for a, b, c in zip(scalar, vector1, vector2):
        assert np.array_equal(a, b)
        assert np.array_equal(a, c)


@pytest.mark.parametrize("size", [(8, 8), (10, 40), (120, 131), (122, 20), (240, 9), (242, 140), (250, 300), (400, 259), (778, 531)])
@pytest.mark.parametrize("distance", [1.0, 2.0])
@pytest.mark.parametrize("layout", ["rgba", "rgb", "gray", "graya"])
def test_filter_kernel_of_two_pixels_per_lane_matches_the_general_one(gpu_decoder, oracle, size, distance, layout):
    """8-bit RGBA / RGB / gray / gray + alpha frames of even width run Gaborish + the first EPF iteration in filter_stream_pairs_kernel
    (two pixels per lane, buffer addressing, mirrored edge pairs loaded from inside the frame); every other layout runs
    filter_stream_kernel.  The same streams through both - widths around the 120-column strips, one and two EPF iterations - must
    give identical bytes for RGBA (both forms convert with v_cvt_pk_u8_f32) and the same pixels up to rounding ties for the other
    layouts (the general form's paths round half up), and match the oracle."""
    w, h = size
    img = synth(w, h, 31 + w)
    src = np.ascontiguousarray({"rgba": img, "rgb": img[..., :3], "gray": img[..., 1:2], "graya": img[..., [1, 3]]}[layout])
    data = oracle.encode(src, distance=distance)
    pairs = gpu_decode(gpu_decoder, [data])[0]
    try:
        assert gpu_decoder.set_option("no_stream_pairs", 1)
        general = gpu_decode(gpu_decoder, [data])[0]
    finally:
        gpu_decoder.set_option("no_stream_pairs", 0)
    assert pairs.shape == general.shape == src.shape
    if layout == "rgba":
        assert np.array_equal(pairs, general)
    else:
        d = np.abs(pairs.astype(np.int32) - general.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3
    check_pixels(pairs, oracle.decode(data).pixels)


@pytest.mark.parametrize("size", [(400, 300), (257, 301), (58, 70)])
@pytest.mark.parametrize("gaborish,epf_iters", [(True, 3), (False, 3), (False, 1), (False, 2)])
def test_streaming_kernels_without_gaborish_and_after_iteration_0(gpu_decoder, oracle, size, gaborish, epf_iters):
    """Every frame with EPF iterations ends in the streaming kernels: frames without Gaborish run them with identity weights, frames with
    three iterations run Gaborish and iteration 0 as stage kernels first.  Against the oracle, against the stage-by-stage decode of the
    same stream (the taps switch the streaming kernels off), and both forms of the streaming kernels against each other."""
    w, h = size
    img = synth(w, h, 17 + epf_iters)
    data = oracle.encode(img, distance=4.5 if epf_iters == 3 else 2.0, epf_iters=epf_iters, gaborish=gaborish)
    ref = oracle.decode(data)
    assert ref.epf_iters == epf_iters
    streamed = gpu_decode(gpu_decoder, [data])[0]
    check_pixels(streamed, ref.pixels)
    staged = gpu_decode(gpu_decoder, [data], taps=True)[0]
    d = np.abs(streamed.astype(np.int32) - staged.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    try:
        assert gpu_decoder.set_option("no_stream_pairs", 1)
        general = gpu_decode(gpu_decoder, [data])[0]
    finally:
        gpu_decoder.set_option("no_stream_pairs", 0)
    assert np.array_equal(streamed, general)


@pytest.mark.parametrize("size", [(400, 300), (777, 531), (257, 300), (56, 64), (57, 9)])
@pytest.mark.parametrize("layout", ["rgba", "rgb", "gray"])
def test_two_epf_iterations_run_as_streaming_kernels(gpu_decoder, oracle, size, layout):
    """Frames with two EPF iterations (distance 1.5 ... 4) decode through filter_stream_kernel (f32 rows out) + filter_stream2_kernel
    instead of three LDS-tiled stage kernels: against the oracle, and against the stage-by-stage decode of the same stream (the taps
    switch the fusion off)."""
    img = synth(size[0], size[1], 9)
    src = np.ascontiguousarray({"rgba": img, "rgb": img[..., :3], "gray": img[..., 1:2]}[layout])
    data = oracle.encode(src, distance=2.5, epf_iters=2)
    ref = oracle.decode(data)
    assert ref.epf_iters == 2
    fused = gpu_decode(gpu_decoder, [data])[0]
    check_pixels(fused, ref.pixels)
    staged = gpu_decode(gpu_decoder, [data], taps=True)[0]
    d = np.abs(fused.astype(np.int32) - staged.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


def test_two_epf_iterations_in_bands(gpu_decoder, oracle):
    """Band decode of a two-iteration frame: the first streaming kernel also writes the row above and below the band for the second."""
    import torch
    from pdn_jpegxl_amd.distributed import decode_frame_band
    img = synth(600, 1500, 10)
    data = oracle.encode(img, distance=2.0)
    assert oracle.decode(data).epf_iters == 2
    whole = gpu_decode(gpu_decoder, [data])[0]
    check_pixels(whole, oracle.decode(data).pixels)
    for world in (2, 3):
        for rank in range(world):
            band, (y0, y1) = decode_frame_band(gpu_decoder, data, rank, world)
            assert np.array_equal(band.cpu().numpy().reshape(y1 - y0, 600, 4), whole[y0:y1]), (world, rank)
