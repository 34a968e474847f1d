"""Deterministic synthetic images for tests and bench.

synth(W, H, seed) -> uint8 (H, W, 4) RGBA: low-frequency cosine gradients + pink noise +
hard-edged rectangles and anti-aliased discs + a soft radial alpha mask in [64, 255].
Generated in 2048x2048 tiles keyed by (seed, tx, ty) so that any band of a large image can
be produced independently (group-row sharding across GPUs needs no full-size host array).
"""
import numpy as np

TILE = 2048


def _pink(rng, h, w, sigma):
    f = rng.standard_normal((h, w)).astype(np.float32)
    F = np.fft.rfft2(f)
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    r = np.sqrt(fy * fy + fx * fx)
    r[0, 0] = 1.0
    F = F / r
    F[0, 0] = 0
    out = np.fft.irfft2(F, s=(h, w)).astype(np.float32)
    out *= sigma / max(out.std(), 1e-6)
    return out


def _tile(seed, tx, ty, W, H, noise_sigma):
    x0, y0 = tx * TILE, ty * TILE
    w, h = min(TILE, W - x0), min(TILE, H - y0)
    rng = np.random.default_rng([seed, tx, ty])
    yy, xx = np.mgrid[y0:y0 + h, x0:x0 + w].astype(np.float32)
    img = np.empty((h, w, 4), np.float32)
    # global (tile-independent) smooth gradients
    for c, (fx, fy, ph) in enumerate([(1.3, 0.7, 0.2), (0.8, 1.1, 1.1), (0.5, 1.7, 2.3)]):
        img[:, :, c] = 128 + 70 * np.cos(2 * np.pi * (fx * xx / max(W, 1) + ph)) * np.cos(2 * np.pi * (fy * yy / max(H, 1)) + ph * 0.5)
    # texture: pink noise, modulated so that part of the tile stays smooth
    mod = 0.5 + 0.5 * np.cos(2 * np.pi * (xx / 900.0)) * np.cos(2 * np.pi * (yy / 700.0))
    mod = np.clip(mod * 1.6 - 0.3, 0, 1)
    for c in range(3):
        img[:, :, c] += _pink(rng, h, w, noise_sigma) * mod
    # hard-edged rectangles
    for _ in range(16):
        rx, ry = rng.integers(0, w), rng.integers(0, h)
        rw, rh = rng.integers(8, max(9, w // 6)), rng.integers(8, max(9, h // 6))
        col = rng.integers(0, 256, 3)
        img[ry:ry + rh, rx:rx + rw, :3] = col
    # anti-aliased discs
    lx, ly = xx - x0, yy - y0
    for _ in range(4):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(10, max(11, min(w, h) / 8))
        d = np.sqrt((lx - cx) ** 2 + (ly - cy) ** 2)
        a = np.clip(r - d + 0.5, 0, 1)[:, :, None]
        col = rng.integers(0, 256, 3).astype(np.float32)
        img[:, :, :3] = img[:, :, :3] * (1 - a) + col * a
    # alpha: soft radial mask in [64, 255]
    d = np.sqrt(((xx - W / 2) / (W / 2)) ** 2 + ((yy - H / 2) / (H / 2)) ** 2)
    img[:, :, 3] = np.clip(255 - 191 * np.clip(d - 0.35, 0, 1) / 0.9, 64, 255)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth(W, H, seed, noise_sigma=6.0, y0=0, y1=None):
    """RGBA8 image rows [y0, y1) of the W x H synthetic image."""
    y1 = H if y1 is None else y1
    out = np.empty((y1 - y0, W, 4), np.uint8)
    for ty in range(y0 // TILE, (y1 - 1) // TILE + 1):
        for tx in range((W + TILE - 1) // TILE):
            t = _tile(seed, tx, ty, W, H, noise_sigma)
            ys, ye = max(y0, ty * TILE), min(y1, ty * TILE + t.shape[0])
            out[ys - y0:ye - y0, tx * TILE:tx * TILE + t.shape[1]] = t[ys - ty * TILE:ye - ty * TILE]
    return out


def synth16(w, h, seed, bits=16):
    """Deeper-sample version of synth(): the 8-bit pattern in the high byte, a deterministic fine pattern below it; samples use
    the low `bits` bits of a uint16 (bits in 9..16)."""
    img = synth(w, h, seed).astype(np.uint32)
    yy, xx = np.mgrid[0:h, 0:w]
    lo = ((xx * 37 + yy * 101 + seed * 13) & 0xFF).astype(np.uint32)[..., None]
    v = (img << 8) | lo
    return (v >> (16 - bits)).astype(np.uint16)
