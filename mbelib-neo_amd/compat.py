"""Per-frame API with the reference's names and argument meaning, for callers and tests that
drive one frame at a time (reference include/mbelib-neo/mbelib.h).  Each call stages one
frame through the HIP launcher (S = T = 1); it is the drop-in surface, not the fast path --
use BatchDecoder for throughput.

State structs are numpy records of ``PARMS_DTYPE`` and are updated in place like the C API
updates ``cur_mp / prev_mp / prev_mp_enhanced``.  The reference's thread-local RNG
(``mbe_setThreadRngSeed``) is a module-level per-"thread" record here.
"""
import numpy as np

from . import _native
from .decoder import ensure_init
from .layout import (
    CODEC_AMBE3600X2450,
    CODEC_IMBE7200X4400,
    FRAME_CELLS,
    PARAM_BITS,
    PARMS_DTYPE,
    RECORD_DTYPE,
    RESULT_DTYPE,
    STATUS_INVALID_ARGUMENT,
    init_parms,
    rng_default,
    rng_seeded,
)

_thread_rng = rng_default(1)


def mbe_setThreadRngSeed(seed):
    global _thread_rng
    _thread_rng = rng_seeded([int(seed) & 0xFFFFFFFF])


def mbe_initMbeParms():
    """Returns (cur_mp, prev_mp, prev_mp_enhanced) with the reference's defaults."""
    p = init_parms(3)
    return p[0:1].copy(), p[1:2].copy(), p[2:3].copy()


def mbe_moveMbeParms(src, dst):
    dst[...] = src


def _pack(codec, fr):
    if fr is None:
        return STATUS_INVALID_ARGUMENT, None
    rows, cols = FRAME_CELLS[codec]
    cells = np.ascontiguousarray(fr, dtype=np.int8).reshape(rows * cols)
    packed = np.zeros(18 if codec == CODEC_IMBE7200X4400 else 9, dtype=np.uint8)
    L = _native.lib()
    fn = L.mbx_pack_imbe7200x4400 if codec == CODEC_IMBE7200X4400 else L.mbx_pack_ambe3600x2450
    return fn(cells.ctypes.data, 1, packed.ctypes.data), packed


def _decode_frame(codec, fr):
    ensure_init(0)
    rc, packed = _pack(codec, fr)
    if rc < 0:
        return rc, None, None
    rec = np.zeros(1, dtype=RECORD_DTYPE)
    _native.check(_native.lib().mbx_fec_host(codec, packed.ctypes.data, 1, rec.ctypes.data), "mbx_fec_host")
    bits = np.zeros(PARAM_BITS[codec], dtype=np.int8)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    _native.lib().mbx_unpack_records(rec.ctypes.data, 1, PARAM_BITS[codec], bits.ctypes.data, res.ctypes.data)
    return int(res["total_errors"][0]), bits, res


def mbe_decodeImbe7200x4400Frame(imbe_fr):
    """-> (ret, imbe_d[88], result).  Negative ret: nothing else is valid."""
    return _decode_frame(CODEC_IMBE7200X4400, imbe_fr)


def mbe_decodeAmbe3600x2450Frame(ambe_fr):
    return _decode_frame(CODEC_AMBE3600X2450, ambe_fr)


def _process_frame(codec, fr, cur_mp, prev_mp, prev_mp_enhanced, want_short):
    global _thread_rng
    ensure_init(0)
    if cur_mp is None or prev_mp is None or prev_mp_enhanced is None:
        return STATUS_INVALID_ARGUMENT, None, None, None
    rc, packed = _pack(codec, fr)
    if rc < 0:
        return rc, None, None, None  # like the reference: no output, no state change
    state = np.zeros(3, dtype=PARMS_DTYPE)
    state[0], state[1], state[2] = cur_mp[0], prev_mp[0], prev_mp_enhanced[0]
    pcm16 = np.zeros(160, dtype=np.int16)
    pcmf = np.zeros(160, dtype=np.float32)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    rec = np.zeros(1, dtype=RECORD_DTYPE)
    rc = _native.lib().mbx_process_batch_host(
        codec, 1, 1, packed.ctypes.data, state.ctypes.data, _thread_rng.ctypes.data, pcm16.ctypes.data,
        pcmf.ctypes.data, res.ctypes.data, rec.ctypes.data,
    )
    _native.check(rc, "mbx_process_batch_host")
    cur_mp[0], prev_mp[0], prev_mp_enhanced[0] = state[0], state[1], state[2]
    bits = np.zeros(PARAM_BITS[codec], dtype=np.int8)
    _native.lib().mbx_unpack_records(rec.ctypes.data, 1, PARAM_BITS[codec], bits.ctypes.data, None)
    return int(res["total_errors"][0]), (pcm16 if want_short else pcmf), res, bits


def mbe_processImbe7200x4400Framef(imbe_fr, cur_mp, prev_mp, prev_mp_enhanced):
    """-> (ret, aout_buf float[160], result, imbe_d[88])"""
    return _process_frame(CODEC_IMBE7200X4400, imbe_fr, cur_mp, prev_mp, prev_mp_enhanced, False)


def mbe_processImbe7200x4400Frame(imbe_fr, cur_mp, prev_mp, prev_mp_enhanced):
    return _process_frame(CODEC_IMBE7200X4400, imbe_fr, cur_mp, prev_mp, prev_mp_enhanced, True)


def mbe_processAmbe3600x2450Framef(ambe_fr, cur_mp, prev_mp, prev_mp_enhanced):
    return _process_frame(CODEC_AMBE3600X2450, ambe_fr, cur_mp, prev_mp, prev_mp_enhanced, False)


def mbe_processAmbe3600x2450Frame(ambe_fr, cur_mp, prev_mp, prev_mp_enhanced):
    return _process_frame(CODEC_AMBE3600X2450, ambe_fr, cur_mp, prev_mp, prev_mp_enhanced, True)


def mbe_synthesizeSpeechf(cur_mp, prev_mp):
    """-> aout_buf float[160]; cur_mp / prev_mp updated in place."""
    global _thread_rng
    ensure_init(0)
    pcmf = np.zeros(160, dtype=np.float32)
    c, p = cur_mp.copy(), prev_mp.copy()
    rc = _native.lib().mbx_synthesize_speech_host(1, c.ctypes.data, p.ctypes.data, _thread_rng.ctypes.data, pcmf.ctypes.data, None)
    _native.check(rc, "mbx_synthesize_speech_host")
    cur_mp[...] = c
    prev_mp[...] = p
    return pcmf


def mbe_floattoshort(float_buf):
    ensure_init(0)
    f = np.ascontiguousarray(float_buf, dtype=np.float32).reshape(1, 160)
    out = np.zeros((1, 160), dtype=np.int16)
    _native.check(_native.lib().mbx_floattoshort_host(f.ctypes.data, out.ctypes.data, 1), "mbx_floattoshort_host")
    return out[0]
