"""Binary layouts shared with the C-ABI (include/mbx_types.h, include/mbx_tables.h) and the
default-state constructors (host logic only, no decoding)."""
import math
import os

import numpy as np

CODEC_IMBE7200X4400 = 0
CODEC_AMBE3600X2450 = 1
CODEC_IMBE7100X4400 = 2  # own FEC / demodulation front end, then the 7200x4400 path
CODEC_AMBE3600X2400 = 3  # D-STAR: the AMBE FEC front end, its own parameter decode and frame policy
FRAME_BYTES = {CODEC_IMBE7200X4400: 18, CODEC_AMBE3600X2450: 9, CODEC_IMBE7100X4400: 18, CODEC_AMBE3600X2400: 9}
PARAM_BITS = {CODEC_IMBE7200X4400: 88, CODEC_AMBE3600X2450: 49, CODEC_IMBE7100X4400: 88, CODEC_AMBE3600X2400: 49}
FRAME_CELLS = {CODEC_IMBE7200X4400: (8, 23), CODEC_AMBE3600X2450: (4, 24), CODEC_IMBE7100X4400: (7, 24), CODEC_AMBE3600X2400: (4, 24)}
ROW_WIDTHS = {
    CODEC_IMBE7200X4400: (23, 23, 23, 23, 15, 15, 15, 7),
    CODEC_AMBE3600X2450: (24, 23, 11, 14),
    CODEC_IMBE7100X4400: (19, 24, 23, 23, 15, 15, 23),
    CODEC_AMBE3600X2400: (24, 23, 11, 14),
}

FLAG_SOFT_INPUT = 0x01
FLAG_C0_VALID = 0x02
FLAG_C4_VALID = 0x04
FLAG_TONE = 0x10
FLAG_ERASURE = 0x20
FLAG_REPEAT = 0x40
FLAG_MUTE = 0x80
STATUS_INVALID_ARGUMENT = -1
STATUS_INVALID_BITS = -2

# struct mbe_parameters -- reference include/mbelib-neo/mbelib.h:88-139 (2604 bytes)
PARMS_DTYPE = np.dtype(
    [
        ("w0", "<f4"),
        ("L", "<i4"),
        ("K", "<i4"),
        ("Vl", "<i4", (57,)),
        ("Ml", "<f4", (57,)),
        ("log2Ml", "<f4", (57,)),
        ("PHIl", "<f4", (57,)),
        ("PSIl", "<f4", (57,)),
        ("gamma", "<f4"),
        ("tonePhase", "<u4"),
        ("swn", "<i4"),
        ("localEnergy", "<f4"),
        ("amplitudeThreshold", "<i4"),
        ("errorRate", "<f4"),
        ("errorCountTotal", "<i4"),
        ("errorCount4", "<i4"),
        ("repeatCount", "<i4"),
        ("mutingThreshold", "<f4"),
        ("previousUw", "<f4", (256,)),
        ("noiseSeed", "<f4"),
        ("noiseOverlap", "<f4", (96,)),
    ]
)
assert PARMS_DTYPE.itemsize == 2604

# mbe_process_result -- reference include/mbelib-neo/mbelib.h:180-191
RESULT_DTYPE = np.dtype(
    [("c0_errors", "<i4"), ("protected_errors", "<i4"), ("c4_errors", "<i4"), ("total_errors", "<i4"), ("flags", "<u4")]
)
assert RESULT_DTYPE.itemsize == 20

RNG_DTYPE = np.dtype(
    [
        ("cn_seed48", "<u8"),
        ("cn_seeded", "<u4"),
        ("unvoiced_seed_state", "<u4"),
        ("unvoiced_seed_override", "<u4"),
        ("reserved", "<u4"),
    ]
)
assert RNG_DTYPE.itemsize == 24

RECORD_DTYPE = np.dtype([("w", "<u4", (4,))])
assert RECORD_DTYPE.itemsize == 16

INT_FIELDS = ("L", "K", "Vl", "tonePhase", "swn", "amplitudeThreshold", "errorCountTotal", "errorCount4", "repeatCount")
FLOAT_FIELDS = ("w0", "Ml", "log2Ml", "PHIl", "PSIl", "gamma", "localEnergy", "errorRate", "mutingThreshold", "previousUw")
EXACT_FLOAT_FIELDS = ("noiseSeed", "noiseOverlap")  # integer-valued floats: must match exactly


def init_parms(n=1):
    """``mbe_initMbeParms`` defaults (reference src/core/mbelib.c:367-410) for n structs."""
    p = np.zeros(n, dtype=PARMS_DTYPE)
    w0 = np.float32((4.0 * math.pi) / (134.0 + 39.5))
    p["w0"] = w0
    p["L"] = int(0.9254 * int((math.pi / float(w0)) + 0.25))
    p["K"] = 12
    p["Ml"] = 1.0
    p["localEnergy"] = 75000.0
    p["amplitudeThreshold"] = 20480
    p["mutingThreshold"] = np.float32(0.0875)
    p["noiseSeed"] = -1.0
    return p


def init_state(streams):
    """State for ``streams`` streams: array [streams, 3] = (cur, prev, prev_enhanced)."""
    return np.repeat(init_parms(1), streams * 3).reshape(streams, 3).copy()


def rng_default(streams):
    r = np.zeros(streams, dtype=RNG_DTYPE)
    r["unvoiced_seed_state"] = 3147
    return r


def rng_seeded(seeds):
    """Per-stream equivalent of ``mbe_setThreadRngSeed`` (reference src/core/mbelib.c:173-181)."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    seeds = np.where(seeds == 0, np.uint64(0x6D25357B), seeds)
    r = np.zeros(seeds.shape[0], dtype=RNG_DTYPE)
    r["cn_seed48"] = (seeds ^ np.uint64(0x5DEECE66D)) & np.uint64((1 << 48) - 1)
    r["cn_seeded"] = 1
    r["unvoiced_seed_state"] = (seeds % np.uint64(53125)).astype(np.uint32)
    r["unvoiced_seed_override"] = 1
    return r


def tables_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mbx_tables.bin")


def load_tables_blob(path=None):
    with open(path or tables_path(), "rb") as f:
        return f.read()


# Offsets inside the table blob that host-side frame generation needs (include/mbx_tables.h).
def table_views(blob):
    """Typed numpy views of the blob members used on the host (encoder side of framegen)."""
    b = np.frombuffer(blob, dtype=np.uint8)
    off = 16
    out = {}

    def take(name, dtype, shape):
        nonlocal off
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        out[name] = b[off : off + n].view(dtype).reshape(shape)
        off += n

    take("golay_matrix", "<u2", (2048,))
    take("golay_gen", "<u2", (12,))
    take("hamming_gen", "<u2", (4,))
    take("hamming_fix", "<u2", (16,))
    take("imbe_w0", "<f4", (208,))
    take("imbe_L", "u1", (208,))
    take("imbe_K", "u1", (208,))
    take("imbe_bo", "u1", (48, 79, 2))
    take("imbe_hoba", "u1", (48, 50))
    take("imbe_ji", "u1", (48, 6))
    take("imbe_ba", "<f4", (48, 5, 2))
    take("imbe_B2", "<f4", (64,))
    take("imbe_quantstep", "<f4", (11,))
    take("imbe_standdev", "<f4", (9,))
    take("imbe_ri_cos", "<f4", (7, 7))
    take("imbe_idct_cos", "<f4", (11, 11, 11))
    take("ambe_w0", "<f4", (120,))
    take("ambe_L", "u1", (120,))
    take("ambe_vuv", "u1", (32, 8))
    take("ambe_lmprbl", "u1", (57, 4))
    take("ambe_dg", "<f4", (32,))
    take("ambe_prba24", "<f4", (512, 3))
    take("ambe_prba58", "<f4", (128, 4))
    take("ambe_hoc_b5", "<f4", (32, 4))
    take("ambe_hoc_b6", "<f4", (16, 4))
    take("ambe_hoc_b7", "<f4", (16, 4))
    take("ambe_hoc_b8", "<f4", (8, 4))
    take("ambe_ri_cos", "<f4", (9, 9))
    take("ambe_idct_cos", "<f4", (18, 18, 18))
    take("ws", "<f4", (321,))
    take("uv_window", "<f4", (256,))
    take("wola_w_prev", "<f4", (160,))
    take("wola_w_curr", "<f4", (160,))
    take("wola_denom", "<f4", (160,))
    take("hamming7100_gen", "<u2", (4,))
    take("hamming7100_fix", "<u2", (16,))
    take("ambep_dg", "<f4", (64,))
    take("ambep_prba24", "<f4", (512, 3))
    take("ambep_prba58", "<f4", (128, 4))
    take("ambep_hoc_b5", "<f4", (16, 4))
    take("ambep_hoc_b6", "<f4", (16, 4))
    take("ambep_hoc_b7", "<f4", (16, 4))
    take("ambep_hoc_b8", "<f4", (16, 4))
    take("ambep_L", "u1", (128,))
    take("ambep_vuv", "u1", (16, 8))
    take("ambep_lmprbl", "u1", (57, 4))
    assert off + 4 == len(b), (off, len(b))
    return out
