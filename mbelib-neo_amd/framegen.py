"""Synthetic channel frames for tests and benches (host side, numpy).

The reference ships no encoder, so clean frames are built here from parameter bits with the
inverse of the FEC stage: Golay(23,12) / Hamming(15,11) systematic encoding from the blob's
generator tables and the same pseudo-random modulation the decoder removes (it is an XOR, so
modulation == demodulation; reference src/imbe/imbe7200x4400.c:636-673,
src/ambe/ambe_common.c:75-100).  Everything is vectorised over frames.
"""
import numpy as np

from .layout import (
    CODEC_AMBE3600X2450,
    CODEC_IMBE7200X4400,
    FRAME_BYTES,
    ROW_WIDTHS,
    load_tables_blob,
    table_views,
)

_SPLITMIX_SEED = 0x9E3779B97F4A7C15

_tables_cache = None


def _tables():
    global _tables_cache
    if _tables_cache is None:
        _tables_cache = table_views(load_tables_blob())
    return _tables_cache


def rng_for(tag):
    """Deterministic generator: PCG64 seeded from the survey's splitmix seed and a tag."""
    return np.random.Generator(np.random.PCG64([_SPLITMIX_SEED & 0xFFFFFFFF, _SPLITMIX_SEED >> 32, int(tag)]))


def random_frames(codec, n, rng):
    """n frames of uniformly random channel bits, packed wire format [n, 18|9] uint8."""
    return rng.integers(0, 256, size=(n, FRAME_BYTES[codec]), dtype=np.uint8)


# ---- systematic encoders ------------------------------------------------------------------

def _parity32(x):
    x = x ^ (x >> 16)
    x = x ^ (x >> 8)
    x = x ^ (x >> 4)
    x = x ^ (x >> 2)
    x = x ^ (x >> 1)
    return x & 1


def golay2312_encode(data12):
    """12-bit data -> 23-bit code word (data in bits 22..11, parity in 10..0)."""
    t = _tables()
    d = np.asarray(data12, dtype=np.uint32)
    par = np.zeros_like(d)
    for i in range(12):
        par ^= np.where((d >> (11 - i)) & 1, np.uint32(t["golay_gen"][i]), np.uint32(0))
    return (d << 11) | par


def hamming1511_encode(data11):
    """11-bit data -> 15-bit code word (data in bits 14..4, parity in 3..0)."""
    t = _tables()
    blk = np.asarray(data11, dtype=np.uint32) << 4
    for m in t["hamming_gen"]:
        m = int(m)
        pbit = m & 0xF  # the single parity bit this check owns
        p = _parity32(blk & np.uint32(m & ~0xF))
        blk = blk | np.where(p == 1, np.uint32(pbit), np.uint32(0))
    return blk


def pr_masks(seed12, widths):
    """XOR masks for rows of the given widths from the sequence x0 = 16*seed, x' = 173x + 13849 mod 2^16."""
    x = (np.asarray(seed12, dtype=np.uint32) * 16) & 0xFFFF
    out = []
    for w in widths:
        m = np.zeros_like(x)
        for _ in range(w):
            x = (173 * x + 13849) & 0xFFFF
            m = (m << 1) | (x >> 15)
        out.append(m)
    return out


def _bits_to_int(bits, lo, hi):
    """bits[:, lo:hi] (MSB first) -> integer array."""
    v = np.zeros(bits.shape[0], dtype=np.uint32)
    for i in range(lo, hi):
        v = (v << 1) | bits[:, i].astype(np.uint32)
    return v


def _pack_rows(rows, widths):
    n = rows[0].shape[0]
    total = sum(widths)
    mat = np.empty((n, total), dtype=np.uint8)
    col = 0
    for r, w in zip(rows, widths):
        for j in range(w - 1, -1, -1):
            mat[:, col] = (r >> j) & 1
            col += 1
    return np.packbits(mat, axis=1)


def encode_imbe7200x4400(param_bits):
    """imbe_d bits [n, 88] (0/1) -> clean wire frames [n, 18]."""
    b = np.asarray(param_bits, dtype=np.uint8)
    u0 = _bits_to_int(b, 0, 12)
    rows = [golay2312_encode(u0)]
    masks = pr_masks(u0, (23, 23, 23, 15, 15, 15))
    for k in range(3):
        rows.append(golay2312_encode(_bits_to_int(b, 12 + 12 * k, 24 + 12 * k)) ^ masks[k])
    for k in range(3):
        rows.append(hamming1511_encode(_bits_to_int(b, 48 + 11 * k, 59 + 11 * k)) ^ masks[3 + k])
    rows.append(_bits_to_int(b, 81, 88))
    return _pack_rows(rows, ROW_WIDTHS[CODEC_IMBE7200X4400])


def encode_ambe3600x2450(param_bits):
    """ambe_d bits [n, 49] (0/1) -> clean wire frames [n, 9]."""
    b = np.asarray(param_bits, dtype=np.uint8)
    u0 = _bits_to_int(b, 0, 12)
    cw0 = golay2312_encode(u0)
    row0 = (cw0 << 1) | _parity32(cw0)  # overall even parity in cell 0
    (m1,) = pr_masks(u0, (23,))
    row1 = golay2312_encode(_bits_to_int(b, 12, 24)) ^ m1
    row2 = _bits_to_int(b, 24, 35)
    row3 = _bits_to_int(b, 35, 49)
    return _pack_rows([row0, row1, row2, row3], ROW_WIDTHS[CODEC_AMBE3600X2450])


def flip_bits(frames, codec, ber, rng):
    """i.i.d. bit flips at rate `ber` over the meaningful channel bits."""
    n, nb = frames.shape
    flips = rng.random((n, nb * 8)) < ber
    return frames ^ np.packbits(flips, axis=1)


# ---- workload generators (BASELINE.json configs) ---------------------------------------------

def imbe_voiced_param_bits(n, rng):
    """Config 2: uniformly random valid b0 in [0, 207], every voicing bit set, the rest random."""
    t = _tables()
    bits = rng.integers(0, 2, size=(n, 88), dtype=np.uint8)
    b0 = rng.integers(0, 208, size=n, dtype=np.int64)
    for k in range(6):
        bits[:, k] = (b0 >> (7 - k)) & 1
    bits[:, 85] = (b0 >> 1) & 1
    bits[:, 86] = b0 & 1
    L9 = t["imbe_L"][b0].astype(np.int64) - 9
    voicing = t["imbe_bo"][L9, :, 0] == 1  # [n, 79]: payload bit i+6 feeds the voicing word
    bits[:, 6:85] |= voicing.astype(np.uint8)
    return bits


def imbe_clean_voiced_frames(n, rng):
    return encode_imbe7200x4400(imbe_voiced_param_bits(n, rng))


def ambe_voice_param_bits(n, rng):
    """Config 3: random voice frames, b0 in [0, 119]; the tone signature is avoided."""
    bits = rng.integers(0, 2, size=(n, 49), dtype=np.uint8)
    b0 = rng.integers(0, 120, size=n, dtype=np.int64)
    bits[:, 0] = (b0 >> 6) & 1
    bits[:, 1] = (b0 >> 5) & 1
    bits[:, 2] = (b0 >> 4) & 1
    bits[:, 3] = (b0 >> 3) & 1
    bits[:, 37] = (b0 >> 2) & 1
    bits[:, 38] = (b0 >> 1) & 1
    bits[:, 39] = b0 & 1
    return bits


def ambe_noisy_voice_frames(n, rng, ber=0.01):
    return flip_bits(encode_ambe3600x2450(ambe_voice_param_bits(n, rng)), CODEC_AMBE3600X2450, ber, rng)


def soft_frames(codec, n, rng, snr_like=2.0):
    """Soft-decision test frames in the reference's array shape, uint8 [n, 184|96, 2] = (bit, reliability):
    random-bit frames observed through additive noise, reliability = clamped |observation| -- so wrong
    hard decisions tend to carry low confidence, plus a share of exact ties (quantised confidences)."""
    cells = {0: 184, 1: 96, 2: 168, 3: 96}[int(codec)]
    bits = rng.integers(0, 2, size=(n, cells), dtype=np.int64)
    obs = (2.0 * bits - 1.0) * snr_like + rng.normal(0.0, 1.0, size=(n, cells))
    hard = (obs > 0).astype(np.uint8)
    rel = np.clip(np.abs(obs) * 40.0, 0, 255).astype(np.uint8)
    coarse = rng.integers(0, 4, size=(n, 1)) == 0          # a quarter of the frames: 3-level confidences
    rel = np.where(coarse, (rel // 96) * 96, rel).astype(np.uint8)
    return np.stack([hard, rel], axis=-1)


def soft_frames_coded(codec, n, rng, snr_like=2.0):
    """Soft-decision frames as a receiver sees them: clean ENCODED voice frames (IMBE 7200x4400 all-voiced / AMBE+2 voice)
    observed through additive noise, reliability = clamped |observation| (snr_like 2.0: about 2.3 % wrong hard decisions,
    low confidence on most of them).  uint8 [n, 184|96, 2] = (bit, reliability), unused cells 0 / 0."""
    from .layout import FRAME_CELLS, ROW_WIDTHS

    codec = int(codec)
    if codec == CODEC_IMBE7200X4400:
        packed = imbe_clean_voiced_frames(n, rng)
    elif codec == CODEC_AMBE3600X2450:
        packed = encode_ambe3600x2450(ambe_voice_param_bits(n, rng))
    else:
        raise ValueError("coded soft frames: IMBE 7200x4400 and AMBE+2 3600x2450 only")
    rows, cols = FRAME_CELLS[codec]
    wire = np.unpackbits(np.ascontiguousarray(packed, dtype=np.uint8), axis=1)
    bits = np.zeros((n, rows, cols), dtype=np.int64)
    used = np.zeros((rows, cols), dtype=bool)
    off = 0
    for r, w in enumerate(ROW_WIDTHS[codec]):
        bits[:, r, :w] = wire[:, off:off + w][:, ::-1]   # the first wire bit of a row is its highest cell
        used[r, :w] = True
        off += w
    bits = bits.reshape(n, rows * cols)
    obs = (2.0 * bits - 1.0) * snr_like + rng.normal(0.0, 1.0, size=bits.shape)
    hard = (obs > 0).astype(np.uint8)
    rel = np.clip(np.abs(obs) * 40.0, 0, 255).astype(np.uint8)
    mask = used.reshape(-1)
    hard[:, ~mask] = 0
    rel[:, ~mask] = 0
    return np.stack([hard, rel], axis=-1)

