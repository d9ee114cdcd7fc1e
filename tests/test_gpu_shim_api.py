"""GPU suite for the per-frame drop-in API (libmbe_neo_amd.so).  These read like the reference's
own CTest programs (tests/test_frame_paths.c, test_input_validation.c, test_floattoshort_parity.c,
test_golden_pcm.c, test_ecc.c, test_params.c) -- same scenarios, same assertions -- with the
library under test swapped for the MI355X one."""
import math
import os

import numpy as np
import pytest

import golden_io
import parity
import shim_lib
from shim_lib import p
from mbelib_neo_amd.layout import PARMS_DTYPE, RESULT_DTYPE

pytestmark = pytest.mark.gpu

AMBE_THR = np.float32(0.096)
IMBE_THR = np.float32(0.0875)


@pytest.fixture(scope="module")
def mbe():
    return shim_lib.load()


def new_state(mbe):
    st = np.zeros(3, dtype=PARMS_DTYPE)
    mbe.mbe_initMbeParms(p(st[0:1]), p(st[1:2]), p(st[2:3]))
    return st[0:1], st[1:2], st[2:3]


def result(total=0, **kw):
    r = np.zeros(1, dtype=RESULT_DTYPE)
    r["total_errors"] = total
    for k, v in kw.items():
        r[k] = v
    return r


def fnv1a32(buf):
    h = 2166136261
    for b in np.ascontiguousarray(buf).view(np.uint8).reshape(-1):
        h = ((h ^ int(b)) * 16777619) & 0xFFFFFFFF
    return h


def set_imbe_b0(d, b0):
    for i in range(8):
        d[i if i < 6 else (85 if i == 6 else 86)] = (b0 >> (7 - i)) & 1


def set_ambe_b0(d, b0):
    for k, i in enumerate((0, 1, 2, 3, 37, 38, 39)):
        d[i] = (b0 >> (6 - k)) & 1


def seed_speech_params(mbe):  # tests/test_params.c:137-150
    cur, prev, dummy = new_state(mbe)
    cur["w0"] = np.float32(0.10)
    cur["L"] = 12
    for l in range(1, 13):
        cur["Vl"][0, l] = 1 if (l % 3) else 0
        cur["Ml"][0, l] = np.float32(0.03) + np.float32(0.001) * np.float32(l)
        cur["PHIl"][0, l] = 0.0
        cur["PSIl"][0, l] = 0.0
    prev[...] = cur
    return cur, prev


# ---- test_api.c ---------------------------------------------------------------------------------
def test_api_basics(mbe):
    assert mbe.mbe_versionString().startswith(b"2.")
    r = result(5)
    mbe.mbe_initProcessResult(p(r))
    assert r.tobytes() == bytes(20)
    mbe.mbe_initProcessResult(None)  # NULL is a no-op


# ---- test_golden_pcm.c --------------------------------------------------------------------------
def test_golden_pcm(mbe):
    g = golden_io.golden_synth()

    def run():
        cur, prev = g["cur_in"].reshape(1).copy(), g["prev_in"].reshape(1).copy()
        out_f = np.zeros(160, dtype=np.float32)
        out_s = np.zeros(160, dtype=np.int16)
        mbe.mbe_setThreadRngSeed(0xC0FFEE)
        mbe.mbe_synthesizeSpeechf(p(out_f), p(cur), p(prev))
        mbe.mbe_floattoshort(p(out_f), p(out_s))
        return out_f, out_s

    f1, s1 = run()
    f2, s2 = run()
    assert f1.tobytes() == f2.tobytes() and s1.tobytes() == s2.tobytes()  # determinism
    m = parity.check_pcm(g["pcmf"], f1, g["pcm16"], s1)
    print("golden pcm via mbe_synthesizeSpeechf:", m, "int16 hash 0x%08X" % fnv1a32(s1))
    assert np.abs(f1).max() < 2e4 and np.abs(s1.astype(np.int32)).max() < 32000  # the reference's sanity bounds


# ---- test_floattoshort_parity.c --------------------------------------------------------------------
def test_floattoshort_parity(mbe):
    def reference_floattoshort(x):  # tests/test_floattoshort_parity.c:20-34
        audio = np.float32(7.0) * np.float32(x)
        if audio != audio:
            audio = np.float32(0.0)
        top = np.float32(32767.0) * np.float32(0.95)
        audio = min(max(audio, -top), top)
        return int(np.float32(audio))  # C cast truncates toward zero

    for case in golden_io.f2s():
        out = np.zeros(160, dtype=np.int16)
        out2 = np.zeros(160, dtype=np.int16)
        mbe.mbe_floattoshort(p(case["inp"].copy()), p(out))
        mbe.mbe_floattoshort(p(case["inp"].copy()), p(out2))
        assert out.tobytes() == out2.tobytes()
        with np.errstate(invalid="ignore", over="ignore"):
            exp = [reference_floattoshort(v) for v in case["inp"]]
        assert list(out) == exp
        assert np.array_equal(out, case["out"])
    mbe.mbe_floattoshort(None, None)  # NULL helpers are no-ops


# ---- test_ecc.c -------------------------------------------------------------------------------------
def test_ecc_known_answers(mbe):
    from mbelib_neo_amd import framegen

    # Hamming(15,11): a code word is a fixed point and every single-bit flip is restored (:186-259)
    rng = np.random.default_rng(3)
    for data in rng.integers(0, 2048, size=12):
        cw = int(framegen.hamming1511_encode(np.array([data]))[0])
        code = np.array([(cw >> j) & 1 for j in range(15)], dtype=np.int8)
        out = np.zeros(15, dtype=np.int8)
        assert mbe.mbe_hamming1511(p(code), p(out)) == 0 and np.array_equal(out, code)
        for j in range(15):
            bad = code.copy()
            bad[j] ^= 1
            assert mbe.mbe_hamming1511(p(bad), p(out)) == 1 and np.array_equal(out, code)
    # Golay: data 0xA55, 1-bit error at data bit 5 -> mbe_checkGolayBlock returns 0xA55 (:356-375)
    cw = int(framegen.golay2312_encode(np.array([0xA55]))[0])
    import ctypes as C

    blk = C.c_long(cw ^ (1 << (11 + 5)))
    assert mbe.mbe_checkGolayBlock(C.byref(blk)) == 0 and blk.value == 0xA55
    # bits above the 23-bit code word pass through the correction like in the reference (src/ecc/ecc.c:246-249)
    hi = C.c_long((0x5A << 23) | (cw ^ (1 << 16)))
    assert mbe.mbe_checkGolayBlock(C.byref(hi)) == 0 and hi.value == ((0x5A << 12) | 0xA55)
    code = np.array([(cw >> j) & 1 for j in range(23)], dtype=np.int8)
    out = np.zeros(23, dtype=np.int8)
    for flips in ((11,), (12, 20), (13, 17, 22)):
        bad = code.copy()
        for j in flips:
            bad[j] ^= 1
        assert mbe.mbe_golay2312(p(bad), p(out)) == len(flips) and np.array_equal(out, code)
    bad = code.copy()
    bad[3] = 2
    assert mbe.mbe_golay2312(p(bad), p(out)) == -2 and mbe.mbe_golay2312(p(code), None) == -1


# ---- test_frame_paths.c -------------------------------------------------------------------------------
def test_frame_paths(mbe):
    for codec, cells_shape, nd, framef, frame, dataf, data, decode in (
        (0, 184, 88, mbe.mbe_processImbe7200x4400Framef, mbe.mbe_processImbe7200x4400Frame, mbe.mbe_processImbe4400Dataf,
         mbe.mbe_processImbe4400Data, mbe.mbe_decodeImbe7200x4400Frame),
        (1, 96, 49, mbe.mbe_processAmbe3600x2450Framef, mbe.mbe_processAmbe3600x2450Frame, mbe.mbe_processAmbe2450Dataf,
         mbe.mbe_processAmbe2450Data, mbe.mbe_decodeAmbe3600x2450Frame),
    ):
        fr = np.zeros(cells_shape, dtype=np.int8)
        fr[[1, 9, 30, 47]] = 1  # a sparse frame
        for want_short in (False, True):
            cur, prev, enh = new_state(mbe)
            out = np.zeros(160, dtype=np.int16 if want_short else np.float32)
            d = np.zeros(nd, dtype=np.int8)
            r = result()
            ret = (frame if want_short else framef)(p(out), p(r), p(fr), p(d), p(cur), p(prev), p(enh))
            assert ret >= 0 and ret == int(r["total_errors"][0]) == int(r["c0_errors"][0] + r["protected_errors"][0])
            assert np.isfinite(out.astype(np.float64)).all() and np.abs(out.astype(np.float64)).max() < 20000
            # decode-only entry agrees, then the Dataf entry with that context gives the same PCM
            d2 = np.zeros(nd, dtype=np.int8)
            r2 = result()
            assert decode(p(fr), p(d2), p(r2)) == ret and np.array_equal(d, d2)
            cur2, prev2, enh2 = new_state(mbe)
            out2 = np.zeros_like(out)
            assert (data if want_short else dataf)(p(out2), p(r2), p(d2), p(cur2), p(prev2), p(enh2)) == ret
            assert out.tobytes() == out2.tobytes() and cur.tobytes() == cur2.tobytes()
            # result may be NULL
            cur3, prev3, enh3 = new_state(mbe)
            out3 = np.zeros_like(out)
            assert (frame if want_short else framef)(p(out3), None, p(fr), p(d), p(cur3), p(prev3), p(enh3)) == ret


# ---- test_input_validation.c ------------------------------------------------------------------------------
def test_input_validation(mbe):
    cur, prev, enh = new_state(mbe)
    before = (cur.tobytes(), prev.tobytes(), enh.tobytes())
    fr = np.zeros(184, dtype=np.int8)
    fr[100] = 2
    out = np.full(160, 123.0, dtype=np.float32)
    d = np.full(88, 9, dtype=np.int8)
    r = result(3)
    assert mbe.mbe_processImbe7200x4400Framef(p(out), p(r), p(fr), p(d), p(cur), p(prev), p(enh)) == -2
    assert (out == 123.0).all() and (d == 9).all() and (cur.tobytes(), prev.tobytes(), enh.tobytes()) == before
    afr = np.zeros(96, dtype=np.int8)
    afr[95] = -1  # an unused cell still counts
    ad = np.full(49, 9, dtype=np.int8)
    assert mbe.mbe_processAmbe3600x2450Framef(p(out), p(r), p(afr), p(ad), p(cur), p(prev), p(enh)) == -2
    assert (out == 123.0).all() and (ad == 9).all()
    fr[100] = 0
    assert mbe.mbe_processImbe7200x4400Framef(None, p(r), p(fr), p(d), p(cur), p(prev), p(enh)) == -1
    assert mbe.mbe_processImbe7200x4400Framef(p(out), p(r), p(fr), None, p(cur), p(prev), p(enh)) == -1
    assert mbe.mbe_processImbe7200x4400Frame(None, p(r), p(fr), p(d), p(cur), p(prev), p(enh)) == -1
    assert mbe.mbe_processImbe4400Dataf(p(out), p(r), p(d * 0), None, p(prev), p(enh)) == -1
    # inconsistent result context -> INVALID_ARGUMENT, nothing written
    d0 = np.zeros(88, dtype=np.int8)
    for bad in (result(3, c0_errors=2, protected_errors=2), result(0, flags=0x100), result(1, c0_errors=-1),
                result(2, c0_errors=3, flags=0x02), result(185)):
        out[:] = 123.0
        assert mbe.mbe_processImbe4400Dataf(p(out), p(bad), p(d0), p(cur), p(prev), p(enh)) == -1
        assert (out == 123.0).all() and (cur.tobytes(), prev.tobytes(), enh.tobytes()) == before
    d0[7] = 5
    assert mbe.mbe_processImbe4400Dataf(p(out), p(result()), p(d0), p(cur), p(prev), p(enh)) == -2
    # L outside [1, 56]: synthesis gives silence, enhancement/smoothing are no-ops
    cur["L"] = 0
    out[:] = 5.0
    mbe.mbe_synthesizeSpeechf(p(out), p(cur), p(prev))
    assert (out == 0.0).all()
    snap = cur.tobytes()
    mbe.mbe_spectralAmpEnhance(p(cur))
    mbe.mbe_applyAdaptiveSmoothing(p(cur), p(prev))
    assert cur.tobytes() == snap
    for fn in (mbe.mbe_spectralAmpEnhance, mbe.mbe_synthesizeComfortNoisef, mbe.mbe_synthesizeSilencef):
        fn(None)
    mbe.mbe_synthesizeSpeechf(None, p(cur), p(prev))
    mbe.mbe_moveMbeParms(None, p(cur))
    assert mbe.mbe_requiresMuting(None) == 0 and mbe.mbe_isMaxFrameRepeat(None) == 0


# ---- test_params.c ------------------------------------------------------------------------------------------
def test_params_repeat_policy_without_c0_context(mbe):
    # :343-395 -- absent C0_VALID the repeat decision depends on total errors only
    for dataf, nd, setb0, total in ((mbe.mbe_processAmbe2450Dataf, 49, set_ambe_b0, 5), (mbe.mbe_processImbe4400Dataf, 88, set_imbe_b0, 11)):
        d = np.zeros(nd, dtype=np.int8)
        setb0(d, 0)
        out = np.zeros(160, dtype=np.float32)
        got = []
        for ctx in (dict(c0_errors=0), dict(c0_errors=4 if nd == 49 else 2, protected_errors=total - (4 if nd == 49 else 2))):
            cur, prev, enh = new_state(mbe)
            r = result(total, **ctx)
            assert dataf(p(out), p(r), p(d), p(cur), p(prev), p(enh)) >= 0
            got.append((int(cur["repeatCount"][0]), bool(r["flags"][0] & 0x40)))
        assert got[0] == got[1] and got[0][1]


def test_params_tone_gate_and_erasure_defaults(mbe):
    # :435-460
    d = np.zeros(49, dtype=np.int8)
    d[0:6] = 1  # tone signature, U3 nibble 0
    set_ambe_b0(d, 120)
    d[4] = d[5] = 1
    out = np.zeros(160, dtype=np.float32)
    cur, prev, enh = new_state(mbe)
    r = result(5)
    assert mbe.mbe_processAmbe2450Dataf(p(out), p(r), p(d), p(cur), p(prev), p(enh)) >= 0 and (r["flags"][0] & 0x10)
    cur, prev, enh = new_state(mbe)
    r = result(6)
    assert mbe.mbe_processAmbe2450Dataf(p(out), p(r), p(d), p(cur), p(prev), p(enh)) >= 0
    assert not (r["flags"][0] & 0x10) and (r["flags"][0] & 0x20)
    assert float(cur["w0"][0]) == 0.0 and int(cur["L"][0]) == 9 and float(prev["w0"][0]) == 0.0 and int(prev["L"][0]) == 9


def test_params_muting_semantics(mbe):
    # :514-549 -- AMBE ignores error-rate muting, IMBE applies it; muted frames still advance localEnergy
    out = np.zeros(160, dtype=np.float32)
    cur, prev = seed_speech_params(mbe)
    cur["mutingThreshold"] = AMBE_THR
    cur["errorRate"] = 1.0
    before = cur["noiseSeed"].tobytes()
    mbe.mbe_synthesizeSpeechf(p(out), p(cur), p(prev))
    assert cur["noiseSeed"].tobytes() != before
    cur, prev = seed_speech_params(mbe)
    cur["mutingThreshold"] = IMBE_THR
    cur["errorRate"] = 1.0
    before, energy = cur["noiseSeed"].tobytes(), cur["localEnergy"].tobytes()
    mbe.mbe_synthesizeSpeechf(p(out), p(cur), p(prev))
    assert cur["noiseSeed"].tobytes() == before and cur["localEnergy"].tobytes() != energy


def test_params_phase_wrap_and_numuv(mbe):
    # :551-571 previous PSI is wrapped to [0, 2pi) before it is advanced
    out = np.zeros(160, dtype=np.float32)
    cur, prev = seed_speech_params(mbe)
    raw = np.float32(20.0 * math.pi) + np.float32(0.321)
    prev["PSIl"][0, 5] = raw
    mbe.mbe_synthesizeSpeechf(p(out), p(cur), p(prev))
    wrapped = math.fmod(float(raw), float(np.float32(2.0 * math.pi)))
    assert abs(float(prev["PSIl"][0, 5]) - wrapped) <= 1e-5
    assert abs(float(cur["PSIl"][0, 5]) - (wrapped + (float(prev["w0"][0]) + float(cur["w0"][0])) * 400.0)) <= 1e-3
    # :620-642 the unvoiced-band count used for the phase jitter includes index 0
    cur, prev, enh = new_state(mbe)
    cur["w0"] = np.float32(0.10)
    cur["L"] = 12
    cur["Vl"][0, :] = 1
    cur["Vl"][0, :13] = 0
    cur["Ml"][0, :] = 0.0
    cur["Ml"][0, :13] = 0.05
    cur["PHIl"][0, :] = 0.0
    cur["PSIl"][0, :] = 0.0
    prev[...] = cur
    mbe.mbe_synthesizeSpeechf(p(out), p(cur), p(prev))
    psi = float(prev["PSIl"][0, 12]) + (float(prev["w0"][0]) + float(cur["w0"][0])) * (12 * 160 / 2.0)
    assert abs(float(cur["PHIl"][0, 12]) - (psi + (13 * -math.pi) / 12)) <= 1e-3


def test_params_tm_seeds_headroom_c4(mbe):
    fx = golden_io.misc_kat()
    # :573-594 Tm is not clamped
    cur, prev, enh = new_state(mbe)
    cur["L"] = 4
    cur["Ml"][0, 1:5] = 10.0
    cur["Vl"][0, 1:5] = 0
    cur["errorRate"] = 0.5
    cur["errorCountTotal"] = 30
    cur["errorCount4"] = 2
    prev["amplitudeThreshold"] = 1
    mbe.mbe_applyAdaptiveSmoothing(p(cur), p(prev))
    assert int(cur["amplitudeThreshold"][0]) == -2999 and float(cur["Ml"][0, 1]) < 0.0
    assert abs(float(cur["Ml"][0, 1]) - float(fx["ml1"])) <= 1e-6 * abs(float(fx["ml1"]))
    # :596-618 seeding drives both generators; cold-start seed = 0x1234 % 53125
    a, b, c = (np.zeros(160, dtype=np.float32) for _ in range(3))
    mbe.mbe_setThreadRngSeed(0x12345678)
    mbe.mbe_synthesizeComfortNoisef(p(a))
    mbe.mbe_setThreadRngSeed(0x12345678)
    mbe.mbe_synthesizeComfortNoisef(p(b))
    mbe.mbe_setThreadRngSeed(0x12340000)
    mbe.mbe_synthesizeComfortNoisef(p(c))
    assert a.tobytes() == b.tobytes() and a.tobytes() != c.tobytes() and np.array_equal(a, fx["comfort"])
    cur, prev = seed_speech_params(mbe)
    cur["noiseSeed"] = -1.0
    cur["noiseOverlap"][0, :] = 0
    mbe.mbe_setThreadRngSeed(0x1234)
    mbe.mbe_synthesizeSpeechf(p(a), p(cur), p(prev))
    assert float(cur["noiseSeed"][0]) == float(0x1234 % 53125)
    # :717-740 repeat headroom reset -> default model
    cur, prev, enh = new_state(mbe)
    prev["repeatCount"] = 4
    cur[...] = prev
    d = np.zeros(88, dtype=np.int8)
    r = result(6)
    mbe.mbe_setThreadRngSeed(77)
    assert mbe.mbe_processImbe4400Dataf(p(a), p(r), p(d), p(cur), p(prev), p(enh)) >= 0
    assert int(cur["repeatCount"][0]) == 0 and int(cur["L"][0]) == 39 and int(cur["Vl"][0, 1]) == 0 and float(cur["Ml"][0, 1]) > 0
    assert abs(float(cur["w0"][0]) - (4.0 * math.pi) / (134.0 + 39.5)) <= 1e-5
    parity.check_pcm(fx["hr_pcm"], a)
    # :644-703 C4 bookkeeping
    for flags, c4, expect in ((0, 0, 0), (0x04, 3, 3), (0x02, 0, 0)):
        cur, prev, enh = new_state(mbe)
        cur["errorCount4"] = 7
        r = result(3 if flags == 0x04 else (1 if flags == 0x02 else 0), flags=flags, c4_errors=c4,
                   protected_errors=3 if flags == 0x04 else 0, c0_errors=1 if flags == 0x02 else 0)
        assert mbe.mbe_processImbe4400Dataf(p(a), p(r), p(d), p(cur), p(prev), p(enh)) >= 0
        assert int(cur["errorCount4"][0]) == expect


def test_streams_frame_by_frame_match_reference_goldens(mbe):
    """The same golden streams as the batch test, but driven one frame at a time through the
    reference-style entry points with per-thread RNG seeding."""
    for codec, fn, nd in ((0, mbe.mbe_processImbe7200x4400Framef, 88), (1, mbe.mbe_processAmbe3600x2450Framef, 49)):
        S, T, fx = golden_io.stream(codec)
        for s in (0, 1, 2, 5):
            cur, prev, enh = new_state(mbe)
            mbe.mbe_setThreadRngSeed(1234 + s)
            got = []
            for t in range(T):
                fr = fx["frames"][s, t]
                out = np.zeros(160, dtype=np.float32)
                d = np.zeros(nd, dtype=np.int8)
                r = result()
                ret = fn(p(out), p(r), p(fr["cells"].copy()), p(d), p(cur), p(prev), p(enh))
                assert ret == int(fr["ret"]) and np.array_equal(d, fr["bits"])
                parity.check_results(fr["result"].reshape(1), r)
                got.append(out)
            parity.check_pcm(fx["frames"][s]["pcmf"], np.array(got))
            st = np.concatenate([cur, prev, enh])
            parity.check_state(fx["final"][s], st)


# ---- soft-decision entry points (reference tests/test_soft_decision.c style: per-frame API) -------
def test_streams_frame_by_frame_through_the_frame_server():
    """The same golden streams with MBE_NEO_FRAME_SERVER=1 (opt-in): synchronous calls served by a resident wavefront from a
    mailbox in pinned memory (mbx_frame_server_start) instead of a launch each -- back-to-back requests, server restarts after
    its idle time-out (a pause between the two codecs), thread-exit shutdown.  Own process: the switch is read once, at load."""
    import subprocess
    import sys

    env = dict(os.environ, MBE_NEO_FRAME_SERVER="1")
    code = ("import time, pytest, sys; sys.path.insert(0, %r); import test_gpu_shim_api as t, shim_lib; m = shim_lib.load(); "
            "t.test_streams_frame_by_frame_match_reference_goldens(m); time.sleep(0.01); t.test_golden_pcm(m); "
            "t.test_ambe2400_entry_points_match_reference_fixture(m); print('served ok')" % os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "served ok" in out.stdout, out.stderr[-2000:]


def _wire_to_cells(wire, widths, shape):
    """wire frames [n, bytes] (rows of `widths` bits, first bit = most significant) -> the per-frame API's cells [n, rows, cols]:
    cell j of a row is bit j of the row's value (ref imbe_fr[8][23] / ambe_fr[4][24])"""
    bits = np.unpackbits(np.ascontiguousarray(wire, dtype=np.uint8), axis=1)
    cells = np.zeros((wire.shape[0],) + shape, dtype=np.int8)
    pos = 0
    for r, w in enumerate(widths):
        cells[:, r, :w] = bits[:, pos:pos + w][:, ::-1]
        pos += w
    return cells


def _voice_cells(name, n, seed):
    from mbelib_neo_amd import framegen
    rng = framegen.rng_for(seed)
    if name == "imbe":
        wire = framegen.flip_bits(framegen.imbe_clean_voiced_frames(n, rng), 0, 0.02, rng)
        return _wire_to_cells(wire, (23, 23, 23, 23, 15, 15, 15, 7), (8, 23))
    return _wire_to_cells(framegen.ambe_noisy_voice_frames(n, rng, ber=0.02), (24, 23, 11, 14), (4, 24))


def _interleaved_channels(mbe, fn, frames, ncell_shape, nd, order):
    """decode `frames` [2, n, ...] of two channels through fn in the given call order; returns PCM [2, n, 160] and final states"""
    st = [new_state(mbe), new_state(mbe)]
    n = frames.shape[1]
    out = np.zeros((2, n, 160), dtype=np.float32)
    pos = [0, 0]
    for ch in order:
        i = pos[ch]
        pos[ch] += 1
        d = np.zeros(nd, dtype=np.int8)
        r = np.zeros(1, dtype=RESULT_DTYPE)
        fr = np.ascontiguousarray(frames[ch, i])
        fn(p(out[ch, i]), p(r), p(fr), p(d), p(st[ch][0]), p(st[ch][1]), p(st[ch][2]))
    return out, st


def test_sync_calls_device_copy_of_the_state_is_invisible(mbe):
    """The synchronous calls keep a device copy of the state and read from it when the caller's structs still are what the
    previous call returned (mbx_process_frame_shadow).  Two channels decoded (a) one after the other -- every call but the
    first finds its structs unchanged: device copy -- and (b) alternating call by call -- every call finds the OTHER channel's
    structs in the pinned block: state taken from the caller -- and (c) with a struct modified by the caller between two calls
    must give bit-identical PCM and states.  HIP vs HIP (the golden-stream tests pin the values)."""
    n = 24
    for name, fn, shape, nd in (("imbe", mbe.mbe_processImbe7200x4400Framef, (8, 23), 88),
                                ("ambe", mbe.mbe_processAmbe3600x2450Framef, (4, 24), 49)):
        cells = _voice_cells(name, 2 * n, 911).reshape(2, n, *shape)
        mbe.mbe_setThreadRngSeed(77)
        a, sa = _interleaved_channels(mbe, fn, cells, shape, nd, [0] * n + [1] * n)
        mbe.mbe_setThreadRngSeed(77)
        b, sb = _interleaved_channels(mbe, fn, cells, shape, nd, [0] * n + [1] * n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
        # (b) alternating: the thread RNG is shared between the channels, so compare against the same order decoded with the
        # device copy switched off in a process of its own
        mbe.mbe_setThreadRngSeed(77)
        c, sc = _interleaved_channels(mbe, fn, cells, shape, nd, [0, 1] * n)
        import subprocess
        import sys
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import test_gpu_shim_api as t, shim_lib; "
                "m = shim_lib.load(); n = %d; name = %r; "
                "fn, shape, nd = ((m.mbe_processImbe7200x4400Framef, (8, 23), 88) if name == 'imbe' else (m.mbe_processAmbe3600x2450Framef, (4, 24), 49)); "
                "cells = t._voice_cells(name, 2 * n, 911).reshape(2, n, *shape); "
                "m.mbe_setThreadRngSeed(77); c, sc = t._interleaved_channels(m, fn, cells, shape, nd, [0, 1] * n); "
                "m.mbe_setThreadRngSeed(77); a, sa = t._interleaved_channels(m, fn, cells, shape, nd, [0] * n + [1] * n); "
                "np.save(sys.argv[1], np.stack([c, a]))" % (os.path.dirname(os.path.abspath(__file__)), n, name))
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "off.npy")
            out = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, MBE_NEO_FRAME_SHADOW="0"), capture_output=True,
                                 text=True, timeout=300)
            assert out.returncode == 0, out.stderr[-2000:]
            off = np.load(path)
        assert np.array_equal(c.view(np.uint32), off[0].view(np.uint32)), name + ": alternating channels"
        assert np.array_equal(a.view(np.uint32), off[1].view(np.uint32)), name + ": channel after channel"
        # (c) the caller changes a struct between two calls: the call after it must see the change
        mbe.mbe_setThreadRngSeed(5)
        st = new_state(mbe)
        outs = []
        for rep in range(2):
            cur, prev, enh = (x.copy() for x in st)
            got = np.zeros((4, 160), dtype=np.float32)
            mbe.mbe_setThreadRngSeed(5)
            for i in range(4):
                d = np.zeros(nd, dtype=np.int8)
                r = np.zeros(1, dtype=RESULT_DTYPE)
                fr = np.ascontiguousarray(cells[0, i])
                if i == 2:
                    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))   # a host that re-initialises its channel
                fn(p(got[i]), p(r), p(fr), p(d), p(cur), p(prev), p(enh))
            outs.append(got)
        fresh = np.zeros((2, 160), dtype=np.float32)
        cur, prev, enh = new_state(mbe)
        mbe.mbe_setThreadRngSeed(5)
        for i in range(4):   # the same calls on a channel that really is new at frame 2 (the thread RNG keeps running)
            d = np.zeros(nd, dtype=np.int8)
            r = np.zeros(1, dtype=RESULT_DTYPE)
            fr = np.ascontiguousarray(cells[0, i])
            if i == 2:
                cur, prev, enh = new_state(mbe)
            tmp_out = np.zeros(160, dtype=np.float32)
            fn(p(tmp_out), p(r), p(fr), p(d), p(cur), p(prev), p(enh))
            if i >= 2:
                fresh[i - 2] = tmp_out
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), name
        assert np.array_equal(outs[0][2:].view(np.uint32), fresh.view(np.uint32)), name + ": re-initialised channel"


def test_soft_entry_points_match_reference_fixture(mbe):
    kat = golden_io.soft_kat()
    for row in kat["golay"][:60]:
        out = np.zeros(23, dtype=np.int8)
        soft = np.ascontiguousarray(row["soft"])
        assert mbe.mbe_golay2312Soft(p(soft), p(out)) == row["ret"]
        assert np.array_equal(out, row["out"])
    for row in kat["hamming"][:60]:
        out = np.zeros(15, dtype=np.int8)
        soft = np.ascontiguousarray(row["soft"])
        assert mbe.mbe_hamming1511Soft(p(soft), p(out)) == row["ret"]
        assert np.array_equal(out, row["out"])
    for name, fn, nbits in (("imbe", mbe.mbe_decodeImbe7200x4400SoftFrame, 88), ("ambe", mbe.mbe_decodeAmbe3600x2450SoftFrame, 49)):
        for row in kat[name][:40]:
            bits = np.zeros(nbits, dtype=np.int8)
            res = np.zeros(1, dtype=RESULT_DTYPE)
            soft = np.ascontiguousarray(row["soft"])
            assert fn(p(soft), p(bits), p(res)) == row["ret"]
            assert np.array_equal(bits, row["bits"])
            assert res[0]["flags"] == row["result"]["flags"] and res[0]["total_errors"] == row["result"]["total_errors"]
    llr = np.ascontiguousarray(kat["llr"]["llr"])
    soft = np.zeros((llr.size, 2), dtype=np.uint8)
    assert mbe.mbe_softBitsFromLlr(p(llr), p(soft), llr.size) == 0
    assert np.array_equal(soft, kat["llr"]["soft"])


def test_soft_frame_rejects_bad_hard_decision(mbe):
    soft = np.zeros((184, 2), dtype=np.uint8)
    soft[17, 0] = 2
    bits = np.zeros(88, dtype=np.int8)
    assert mbe.mbe_decodeImbe7200x4400SoftFrame(p(soft), p(bits), None) == -2   # MBE_STATUS_INVALID_BITS
    assert mbe.mbe_decodeImbe7200x4400SoftFrame(p(soft), None, None) == -1


def test_soft_process_stream_matches_reference(mbe):
    """12 frames through mbe_processImbe7200x4400SoftFramef, one stream, seed 4242 (reference fixture)."""
    kat = golden_io.soft_kat()["process"]
    cur, prev, enh = (np.zeros(1, dtype=PARMS_DTYPE) for _ in range(3))
    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
    mbe.mbe_setThreadRngSeed(4242)
    pcm = np.zeros((len(kat), 160), dtype=np.float32)
    for t, row in enumerate(kat):
        bits = np.zeros(88, dtype=np.int8)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        soft = np.ascontiguousarray(row["soft"])
        ret = mbe.mbe_processImbe7200x4400SoftFramef(p(pcm[t]), p(res), p(soft), p(bits), p(cur), p(prev), p(enh))
        assert ret == row["ret"]
        assert res[0]["flags"] == row["result"]["flags"]
    parity.check_pcm(kat["pcmf"], pcm)


# ---- IMBE 7100x4400 entry points ------------------------------------------------------------------
def test_imbe7100_entry_points_match_reference_fixture(mbe):
    kat = golden_io.imbe7100_kat()
    for row in kat["hamming"][::257]:
        cells = np.array([(int(row["inp"]) >> j) & 1 for j in range(15)], dtype=np.int8)
        out = np.zeros(15, dtype=np.int8)
        assert mbe.mbe_7100x4400hamming1511(p(cells), p(out)) == row["errs"]
        assert sum(int(out[j]) << j for j in range(15)) == int(row["out"])
    for row in kat["fec"][:48]:
        bits = np.zeros(88, dtype=np.int8)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        cells = np.ascontiguousarray(row["cells"])
        assert mbe.mbe_decodeImbe7100x4400Frame(p(cells), p(bits), p(res)) == row["ret"]
        assert np.array_equal(bits, row["bits"])
        assert res[0]["flags"] == row["result"]["flags"] and res[0]["c4_errors"] == row["result"]["c4_errors"]
    st = kat["stream"][0]
    cur, prev, enh = (np.zeros(1, dtype=PARMS_DTYPE) for _ in range(3))
    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
    mbe.mbe_setThreadRngSeed(1234)
    T = len(st["frames"])
    pcm = np.zeros((T, 160), dtype=np.float32)
    for t, row in enumerate(st["frames"]):
        bits = np.zeros(88, dtype=np.int8)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        cells = np.ascontiguousarray(row["cells"])
        assert mbe.mbe_processImbe7100x4400Framef(p(pcm[t]), p(res), p(cells), p(bits), p(cur), p(prev), p(enh)) == row["ret"]
        assert res[0]["flags"] == row["result"]["flags"]
    parity.check_pcm(st["frames"]["pcmf"], pcm)
    parity.check_state(st["final"].reshape(1), cur)


def test_imbe7100_soft_entry_points(mbe):
    kat = golden_io.imbe7100_kat()
    for row in kat["hamming_soft"][:50]:
        out = np.zeros(15, dtype=np.int8)
        soft = np.ascontiguousarray(row["soft"])
        assert mbe.mbe_7100x4400hamming1511Soft(p(soft), p(out)) == row["ret"]
        assert np.array_equal(out, row["out"])
    for row in kat["fec_soft"][:40]:
        bits = np.zeros(88, dtype=np.int8)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        soft = np.ascontiguousarray(row["soft"])
        assert mbe.mbe_decodeImbe7100x4400SoftFrame(p(soft), p(bits), p(res)) == row["ret"]
        assert np.array_equal(bits, row["bits"]) and res[0]["flags"] == row["result"]["flags"]


# ---- AMBE 3600x2400 / D-STAR entry points ---------------------------------------------------------------
def test_ambe2400_entry_points_match_reference_fixture(mbe):
    framed, data = golden_io.ambe2400_kat()
    for s in (0, 3):   # a random-bit stream and a mostly clean one, frame by frame
        st = framed[s]
        cur, prev, enh = (np.zeros(1, dtype=PARMS_DTYPE) for _ in range(3))
        mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
        mbe.mbe_setThreadRngSeed(1234 + s)
        T = len(st["frames"])
        pcm = np.zeros((T, 160), dtype=np.float32)
        for t, row in enumerate(st["frames"]):
            bits = np.zeros(49, dtype=np.int8)
            res = np.zeros(1, dtype=RESULT_DTYPE)
            cells = np.ascontiguousarray(row["cells"])
            assert mbe.mbe_processAmbe3600x2400Framef(p(pcm[t]), p(res), p(cells), p(bits), p(cur), p(prev), p(enh)) == row["ret"]
            assert np.array_equal(bits, row["bits"]) and res[0]["flags"] == row["result"]["flags"]
        parity.check_pcm(st["frames"]["pcmf"], pcm)
        parity.check_state(st["final"].reshape(1, 3), np.concatenate([cur, prev, enh]).reshape(1, 3))
    st = data[1]   # scripted parameter-bit stream: tones, silence classes, repeats
    cur, prev, enh = (np.zeros(1, dtype=PARMS_DTYPE) for _ in range(3))
    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
    mbe.mbe_setThreadRngSeed(5001)
    pcm = np.zeros((len(st["frames"]), 160), dtype=np.float32)
    for t, row in enumerate(st["frames"]):
        res = np.zeros(1, dtype=RESULT_DTYPE)
        res[0]["total_errors"] = row["total_in"]
        bits = np.ascontiguousarray(row["bits"])
        assert mbe.mbe_processAmbe2400Dataf(p(pcm[t]), p(res), p(bits), p(cur), p(prev), p(enh)) == row["ret"]
        assert res[0]["flags"] == row["result"]["flags"]
    parity.check_pcm(st["frames"]["pcmf"], pcm)
    parity.check_state(st["final"].reshape(1, 3), np.concatenate([cur, prev, enh]).reshape(1, 3))


def test_tone_and_format_entry_points(mbe):
    import ctypes as C

    ambe, dstar = golden_io.tone_kat()
    cur, prev, enh = (np.zeros(1, dtype=PARMS_DTYPE) for _ in range(3))
    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
    for row in ambe:
        pcm = np.zeros(160, dtype=np.float32)
        bits = np.ascontiguousarray(row["bits"])
        mbe.mbe_synthesizeTonef(p(pcm), p(bits), p(cur))
        assert np.max(np.abs(pcm - row["pcmf"])) <= 2e-3   # sinf of the device libm vs glibc on a 4400-peak tone
        assert int(cur["swn"][0]) == row["swn"] and int(cur["tonePhase"][0]) == row["tonePhase"]
    mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
    for row in dstar:
        pcm = np.zeros(160, dtype=np.float32)
        mbe.mbe_synthesizeTonefdstar(p(pcm), None, p(cur), int(row["id"]))
        assert np.max(np.abs(pcm - row["pcmf"])) <= 2e-3
        assert int(cur["swn"][0]) == row["swn"]
    res = np.zeros(1, dtype=RESULT_DTYPE)
    res[0]["total_errors"] = 3
    res[0]["flags"] = 0x20 | 0x40
    buf = C.create_string_buffer(16)
    mbe.mbe_formatProcessResult(buf, 16, p(res))
    assert buf.value == b"===ER"
    mbe.mbe_formatProcessResult(buf, 4, p(res))
    assert buf.value == b"==="


# ---- re-entrancy: ref include/mbelib-neo/mbelib.h:28-30 (re-entrant per stream, helper state thread-local) --------
def _decode_streams(mbe, codec, streams, fx, T):
    """what one host thread does: its streams one after the other, frame by frame, thread-local RNG seeded per stream"""
    fn, nd = ((mbe.mbe_processImbe7200x4400Framef, 88), (mbe.mbe_processAmbe3600x2450Framef, 49))[codec]
    out = {}
    for s in streams:
        cur, prev, enh = new_state(mbe)
        mbe.mbe_setThreadRngSeed(1234 + s)
        pcm = np.zeros((T, 160), dtype=np.float32)
        rets = []
        for t in range(T):
            fr = fx["frames"][s, t]
            d = np.zeros(nd, dtype=np.int8)
            r = result()
            cells = fr["cells"].copy()   # held in a local: a temporary would be freed (and reused by another thread) before the call reads it
            rets.append((fn(p(pcm[t]), p(r), p(cells), p(d), p(cur), p(prev), p(enh)), r.tobytes(), d.tobytes()))
        out[s] = (pcm, rets, np.concatenate([cur, prev, enh]))
    return out


@pytest.mark.parametrize("codec", [0, 1])
def test_four_host_threads_decode_concurrently(mbe, codec):
    """4 host threads x 4 streams x 24 frames through libmbe_neo_amd.so at the same time (ctypes releases the GIL
    around every call).  Each thread must get, bit for bit, what a single thread gets for the same streams -- the
    threads share nothing but the device -- and stay within tolerance of the reference's golden streams."""
    import threading

    S, T, fx = golden_io.stream(codec)
    T = min(T, 24)
    groups = [list(range(4 * k, 4 * k + 4)) for k in range(4)]
    serial = {}
    for g in groups:
        serial.update(_decode_streams(mbe, codec, g, fx, T))
    for attempt in range(3):   # three concurrent rounds: a race needs the interleaving to happen
        got, errors = {}, []

        def work(g):
            try:
                got.update(_decode_streams(mbe, codec, g, fx, T))
            except Exception as e:   # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(g,)) for g in groups]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for s in sorted(serial):
            pcm, rets, st = got[s]
            assert pcm.tobytes() == serial[s][0].tobytes(), f"stream {s}: PCM differs from the single-threaded run"
            assert rets == serial[s][1] and st.tobytes() == serial[s][2].tobytes()
    for s in sorted(serial):
        pcm, rets, st = serial[s]
        assert [r[0] for r in rets] == [int(x) for x in fx["frames"][s, :T]["ret"]]
        parity.check_pcm(fx["frames"][s, :T]["pcmf"], pcm)


# ---- the classic call sequence, stage by stage (mbelib.h:286-307, 381-387, 457-463, 531-537) ------------------------
def test_stage_helpers_reproduce_the_frame_decode(mbe):
    """ecc C0 -> demodulate -> ecc data, called one by one on the reference's own fixture frames, must give the
    parameter bits and error counts the reference's frame decode gave (fixtures made by the real reference); cells
    outside the wire rows stay untouched; invalid cells are rejected like in the reference."""
    kat7100 = golden_io.imbe7100_kat()
    cases = (
        (golden_io.fec(0)[:200], 88, (8, 23), mbe.mbe_eccImbe7200x4400C0, mbe.mbe_demodulateImbe7200x4400Data, mbe.mbe_eccImbe7200x4400Data, None),
        (golden_io.fec(1)[:200], 49, (4, 24), mbe.mbe_eccAmbe3600x2450C0, mbe.mbe_demodulateAmbe3600x2450Data, mbe.mbe_eccAmbe3600x2450Data, None),
        (golden_io.fec(1)[200:300], 49, (4, 24), mbe.mbe_eccAmbe3600x2400C0, mbe.mbe_demodulateAmbe3600x2400Data, mbe.mbe_eccAmbe3600x2400Data, None),
        (kat7100["fec"][:200], 88, (7, 24), mbe.mbe_eccImbe7100x4400C0, mbe.mbe_demodulateImbe7100x4400Data, mbe.mbe_eccImbe7100x4400Data,
         mbe.mbe_convertImbe7100to7200),
    )
    widths = {(8, 23): (23, 23, 23, 23, 15, 15, 15, 7), (4, 24): (24, 23, 11, 14), (7, 24): (19, 24, 23, 23, 15, 15, 23)}
    for fx, nbits, shape, ecc_c0, demod, ecc_data, convert in cases:
        for row in fx:
            cells = row["cells"].copy()
            before = cells.copy()
            c0 = ecc_c0(p(cells))
            assert demod(p(cells)) == 0
            bits = np.zeros(nbits, dtype=np.int8)
            prot = ecc_data(p(cells), p(bits))
            if convert is not None:
                assert convert(p(bits)) == 0
            assert np.array_equal(bits, row["bits"])
            assert c0 == row["result"]["c0_errors"] and prot == row["result"]["protected_errors"]
            grid, was = cells.reshape(shape), before.reshape(shape)
            for r, w in enumerate(widths[shape]):   # cells that are not on the wire are never written
                assert np.array_equal(grid[r, w:], was[r, w:])
        bad = fx[0]["cells"].copy()
        bad[-1] = 3   # even an unused cell is validated (tests/test_input_validation.c)
        keep = bad.copy()
        assert ecc_c0(p(bad)) == -2 and demod(p(bad)) == -2 and ecc_data(p(bad), p(bits)) == -2 and np.array_equal(bad, keep)
        assert ecc_data(p(fx[0]["cells"].copy()), None) == -1
    for row in kat7100["convert"]:
        d = row["inp"].copy()
        assert mbe.mbe_convertImbe7100to7200(p(d)) == 0 and np.array_equal(d, row["out"])
    for fn in (mbe.mbe_dumpImbe4400Data, mbe.mbe_dumpAmbe2450Data, mbe.mbe_dumpImbe7200x4400Frame):
        fn(None)   # NULL dumps print nothing and do not crash


def test_decode_parms_known_answers(mbe):
    """tests/golden/params_kat.bin (the real reference): every b0 through mbe_decodeImbe4400Parms / mbe_decodeAmbe2450Parms
    (return code, w0, L, K) and 64 full parameter decodes per codec against a patterned prev_mp (tests/test_params.c)."""
    t_imbe, t_ambe, full_imbe, full_ambe = golden_io.params_kat()
    for b0 in range(256):
        d = np.zeros(88, dtype=np.int8)
        set_imbe_b0(d, b0)
        cur, prev, _ = new_state(mbe)
        rc = mbe.mbe_decodeImbe4400Parms(p(d), p(cur), p(prev))
        r = t_imbe[b0]
        assert rc == r["rc"], b0
        if rc == 0:
            assert cur["w0"][0] == r["w0"] and cur["L"][0] == r["L"] and cur["K"][0] == r["K"]
    for b0 in range(128):
        d = np.zeros(49, dtype=np.int8)
        set_ambe_b0(d, b0)
        cur, prev, _ = new_state(mbe)
        rc = mbe.mbe_decodeAmbe2450Parms(p(d), p(cur), p(prev))
        r = t_ambe[b0]
        assert rc == r["rc"], b0
        if rc == 0:
            assert cur["w0"][0] == r["w0"] and cur["L"][0] == r["L"]
    for fn, fx in ((mbe.mbe_decodeImbe4400Parms, full_imbe), (mbe.mbe_decodeAmbe2450Parms, full_ambe)):
        for r in fx:
            cur, prev, _ = new_state(mbe)
            l = np.arange(57)
            prev["log2Ml"][0] = (np.float32(0.25) * ((l * 7) % 11).astype(np.float32) - np.float32(1.0)).astype(np.float32)
            prev["Ml"][0] = np.exp2(prev["log2Ml"][0].astype(np.float32)).astype(np.float32)
            prev["L"] = r["prev_L"]
            prev["gamma"] = 1.5
            rc = fn(p(np.ascontiguousarray(r["bits"])), p(cur), p(prev))
            assert rc == r["rc"]
            if rc == 0:
                parity.check_state(r["cur"].reshape(1), cur)
    assert mbe.mbe_decodeImbe4400Parms(p(np.zeros(88, dtype=np.int8)), None, None) == -1
    two = np.zeros(88, dtype=np.int8)
    two[3] = 2
    cur, prev, _ = new_state(mbe)
    assert mbe.mbe_decodeImbe4400Parms(p(two), p(cur), p(prev)) == -2


# ---- queue mode: the per-frame API fanning frames into batched launches (include/mbe_neo_amd.h) ---------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_queue_mode_matches_reference_goldens(mbe, mode):
    """All 64 golden IMBE streams and all 64 golden AMBE+2 streams decoded tick by tick through the queued per-frame
    calls -- both codecs in the same flush, int16 and float outputs mixed -- in write-back (0) and resident (1) state
    mode.  Everything the synchronous calls deliver must be there after mbe_flush(): return-value-in-result, parameter
    bits, flags, PCM within tolerance, and (after the flush / after mbe_batchEnd) the three structs."""
    Si, Ti, fi = golden_io.stream(0)
    Sa, Ta, fa = golden_io.stream(1)
    T = min(Ti, Ta, 12)
    chans = []
    for codec, S, fx, fn_f, fn_s, nd in ((0, Si, fi, mbe.mbe_processImbe7200x4400Framef, mbe.mbe_processImbe7200x4400Frame, 88),
                                         (1, Sa, fa, mbe.mbe_processAmbe3600x2450Framef, mbe.mbe_processAmbe3600x2450Frame, 49)):
        for s in range(S):
            cur, prev, enh = new_state(mbe)
            short = (s % 3) == 0
            chans.append(dict(codec=codec, s=s, fx=fx, fn=fn_s if short else fn_f, short=short, cur=cur, prev=prev, enh=enh,
                              pcm=np.zeros((T, 160), dtype=np.int16 if short else np.float32), res=np.zeros(T, dtype=RESULT_DTYPE),
                              bits=np.zeros((T, nd), dtype=np.int8), cells=[None] * T))
    assert mbe.mbe_batchBegin(mode) == 0 and mbe.mbe_batchBegin(mode) == -1
    for t in range(T):
        for ch in chans:
            if t == 0:
                mbe.mbe_setThreadRngSeed(1234 + ch["s"])   # a channel takes the thread's RNG state at its first queued frame
            ch["cells"][t] = ch["fx"]["frames"][ch["s"], t]["cells"].copy()
            assert ch["fn"](p(ch["pcm"][t]), p(ch["res"][t:t + 1]), p(ch["cells"][t]), p(ch["bits"][t]), p(ch["cur"]), p(ch["prev"]),
                            p(ch["enh"])) == 0
        assert mbe.mbe_batchPending() == len(chans)
        assert mbe.mbe_flush() == len(chans) and mbe.mbe_batchPending() == 0
    bad = chans[0]["cells"][0].copy()
    bad[5] = 7
    assert chans[0]["fn"](p(chans[0]["pcm"][0]), None, p(bad), p(chans[0]["bits"][0]), p(chans[0]["cur"]), p(chans[0]["prev"]),
                          p(chans[0]["enh"])) == -2 and mbe.mbe_batchPending() == 0
    assert mbe.mbe_batchEnd() == 0 and mbe.mbe_batchEnd() == -1
    for ch in chans:
        fr = ch["fx"]["frames"][ch["s"], :T]
        assert np.array_equal(ch["bits"], fr["bits"])
        parity.check_results(fr["result"], ch["res"])
        assert np.array_equal(ch["res"]["total_errors"], fr["ret"])
        if ch["short"]:
            d = np.abs(ch["pcm"].astype(np.int32) - fr["pcm16"].astype(np.int32))
            assert d.max() <= 3 and np.mean(d <= 1) >= 0.999
        else:
            parity.check_pcm(fr["pcmf"], ch["pcm"])
    # the structs came home: continue two of the streams synchronously and land on the reference's final state
    for ch in (chans[1], chans[Si + 2]):
        Tfull = Ti if ch["codec"] == 0 else Ta
        fn = mbe.mbe_processImbe7200x4400Framef if ch["codec"] == 0 else mbe.mbe_processAmbe3600x2450Framef
        if mode == 0:
            pass   # write-back mode dropped the per-channel RNG at every flush: only the model state is compared below
        for t in range(T, Tfull):
            out = np.zeros(160, dtype=np.float32)
            d = np.zeros(ch["bits"].shape[1], dtype=np.int8)
            cells = ch["fx"]["frames"][ch["s"], t]["cells"].copy()
            assert fn(p(out), None, p(cells), p(d), p(ch["cur"]), p(ch["prev"]), p(ch["enh"])) == int(ch["fx"]["frames"][ch["s"], t]["ret"])
        st = np.concatenate([ch["cur"], ch["prev"], ch["enh"]])
        for name in ("L", "K", "Vl", "repeatCount", "errorCountTotal", "errorCount4", "amplitudeThreshold"):
            assert np.array_equal(ch["fx"]["final"][ch["s"]][name], st[name]), name


def test_queue_mode_ragged_and_direct_calls(mbe):
    """Channels with different numbers of queued frames in one flush (1, 2 and 3), a synchronous call on a resident
    channel in the middle (flushes and releases it), and mbe_batchRelease: each channel's PCM sequence equals the one
    the synchronous API produces for it."""
    S, T, fx = golden_io.stream(0)
    T = min(T, 9)
    fn = mbe.mbe_processImbe7200x4400Framef

    def sync_run(s):
        cur, prev, enh = new_state(mbe)
        mbe.mbe_setThreadRngSeed(1234 + s)
        pcm = np.zeros((T, 160), dtype=np.float32)
        for t in range(T):
            cells = fx["frames"][s, t]["cells"].copy()
            d = np.zeros(88, dtype=np.int8)
            fn(p(pcm[t]), None, p(cells), p(d), p(cur), p(prev), p(enh))
        return pcm, np.concatenate([cur, prev, enh])

    want = {s: sync_run(s) for s in range(6)}
    st = {s: new_state(mbe) for s in range(6)}
    pcm = {s: np.zeros((T, 160), dtype=np.float32) for s in range(6)}
    done = {s: 0 for s in range(6)}
    keep = []
    assert mbe.mbe_batchBegin(1) == 0
    plan = [(1, 2, 3, 1, 2, 3), (3, 1, 1, 2, 2, 1), (2, 3, 2, 3, 1, 2), (3, 3, 3, 3, 4, 3)]
    for k, counts in enumerate(plan):
        for s, cnt in enumerate(counts):
            for _ in range(cnt):
                t = done[s]
                if t >= T:
                    continue
                if t == 0:
                    mbe.mbe_setThreadRngSeed(1234 + s)
                cells = fx["frames"][s, t]["cells"].copy()
                d = np.zeros(88, dtype=np.int8)
                keep.append((cells, d))
                if k == 2 and s == 4:   # a synchronous call (soft-decision entry point on hard bits would differ; use the Data call)
                    mbe.mbe_decodeImbe7200x4400Frame(p(cells), p(d), None)
                    r = result()
                    mbe.mbe_decodeImbe7200x4400Frame(p(cells), p(d), p(r))
                    assert mbe.mbe_processImbe4400Dataf(p(pcm[s][t]), p(r), p(d), p(st[s][0]), p(st[s][1]), p(st[s][2])) >= 0
                else:
                    assert fn(p(pcm[s][t]), None, p(cells), p(d), p(st[s][0]), p(st[s][1]), p(st[s][2])) == 0
                done[s] += 1
        assert mbe.mbe_flush() >= 0
        if k == 1:
            assert mbe.mbe_batchRelease(p(st[0][0])) == 0   # channel 0 leaves the pool and comes back at its next frame
    assert mbe.mbe_batchEnd() >= 0
    for s in range(6):
        assert done[s] == T
        if s == 4:
            continue   # took a fresh copy of the thread RNG state after the synchronous call: only compared where no noise is involved
        assert pcm[s].tobytes() == want[s][0].tobytes(), f"channel {s}"
        assert np.concatenate(st[s]).tobytes() == want[s][1].tobytes()


def test_queue_mode_other_codecs_equal_synchronous_calls(mbe):
    """IMBE 7100x4400 and AMBE 3600x2400 channels through the queue (both in one flush, three frames per channel and
    flush): PCM, results, parameter bits and the final structs are bit for bit those of the synchronous calls on the same
    frames (reference fixtures imbe7100_kat.bin / ambe2400_kat.bin)."""
    k7100 = golden_io.imbe7100_kat()["stream"]
    k2400 = golden_io.ambe2400_kat()[0]
    plans = []
    for s in range(4):
        plans.append((mbe.mbe_processImbe7100x4400Framef, 88, k7100[s]["frames"]["cells"], 1234 + s))
        plans.append((mbe.mbe_processAmbe3600x2400Framef, 49, k2400[s]["frames"]["cells"], 1234 + s))

    def run(queued):
        outs = []
        chans = []
        for fn, nd, cells, seed in plans:
            T = len(cells)
            cur, prev, enh = new_state(mbe)
            chans.append(dict(fn=fn, nd=nd, cells=[c.copy() for c in cells], seed=seed, cur=cur, prev=prev, enh=enh, T=T,
                              pcm=np.zeros((T, 160), dtype=np.float32), res=np.zeros(T, dtype=RESULT_DTYPE),
                              bits=np.zeros((T, nd), dtype=np.int8), rets=[]))
        if queued:
            assert mbe.mbe_batchBegin(1) == 0
        tmax = max(c["T"] for c in chans)
        for t0 in range(0, tmax, 3):
            for ch in chans:
                for t in range(t0, min(t0 + 3, ch["T"])):
                    if t == 0:
                        mbe.mbe_setThreadRngSeed(ch["seed"])
                    elif not queued:
                        pass
                    ch["rets"].append(ch["fn"](p(ch["pcm"][t]), p(ch["res"][t:t + 1]), p(ch["cells"][t]), p(ch["bits"][t]), p(ch["cur"]),
                                               p(ch["prev"]), p(ch["enh"])))
            if queued:
                assert mbe.mbe_flush() >= 0
        if queued:
            assert mbe.mbe_batchEnd() >= 0
        return chans

    # synchronous: one channel after the other would share the thread RNG across channels; run each channel alone
    sync = []
    for plan in plans:
        fn, nd, cells, seed = plan
        T = len(cells)
        cur, prev, enh = new_state(mbe)
        mbe.mbe_setThreadRngSeed(seed)
        pcm = np.zeros((T, 160), dtype=np.float32)
        res = np.zeros(T, dtype=RESULT_DTYPE)
        bits = np.zeros((T, nd), dtype=np.int8)
        for t in range(T):
            c = cells[t].copy()
            assert fn(p(pcm[t]), p(res[t:t + 1]), p(c), p(bits[t]), p(cur), p(prev), p(enh)) == res[t]["total_errors"]
        sync.append((pcm, res, bits, np.concatenate([cur, prev, enh])))
    got = run(True)
    for ch, (pcm, res, bits, st) in zip(got, sync):
        assert all(r == 0 for r in ch["rets"])
        assert ch["pcm"].tobytes() == pcm.tobytes() and ch["res"].tobytes() == res.tobytes() and np.array_equal(ch["bits"], bits)
        assert np.concatenate([ch["cur"], ch["prev"], ch["enh"]]).tobytes() == st.tobytes()


def test_queue_mode_resident_channel_reset_and_copied_by_direct_calls(mbe):
    """A host that resets a resident channel with mbe_initMbeParms, or copies one of its structs with mbe_moveMbeParms,
    works on the CURRENT state: the channel is flushed and released first (before round 3 the reset was silently overwritten
    by the device state at mbe_batchEnd, and the copy returned the stale host struct)."""
    S, T, fx = golden_io.stream(0)
    fn = mbe.mbe_processImbe7200x4400Framef

    def frame(s, t):
        return fx["frames"][s, t]["cells"].copy(), np.zeros(88, dtype=np.int8)

    # reference behaviour by synchronous calls: three frames, reset, three frames again
    cur, prev, enh = new_state(mbe)
    mbe.mbe_setThreadRngSeed(77)
    want = np.zeros((6, 160), dtype=np.float32)
    snap = None
    for t in range(6):
        if t == 3:
            snap = np.concatenate([cur, prev, enh]).copy()
            mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))
            mbe.mbe_setThreadRngSeed(78)
        cells, d = frame(0, t)
        fn(p(want[t]), None, p(cells), p(d), p(cur), p(prev), p(enh))
    want_state = np.concatenate([cur, prev, enh]).copy()

    cur, prev, enh = new_state(mbe)
    copy_of_prev = new_state(mbe)[0]
    got = np.zeros((6, 160), dtype=np.float32)
    keep = []
    mbe.mbe_setThreadRngSeed(77)
    assert mbe.mbe_batchBegin(1) == 0   # resident: the state stays on the device between flushes
    for t in range(6):
        if t == 3:
            mbe.mbe_moveMbeParms(p(prev), p(copy_of_prev))          # reads a struct of a resident channel
            assert copy_of_prev.tobytes() == snap[1:2].tobytes()
            mbe.mbe_initMbeParms(p(cur), p(prev), p(enh))           # resets it
            mbe.mbe_setThreadRngSeed(78)                            # (a channel copies the thread's RNG state at its first queued frame)
        cells, d = frame(0, t)
        keep.append((cells, d))
        assert fn(p(got[t]), None, p(cells), p(d), p(cur), p(prev), p(enh)) == 0
        assert mbe.mbe_flush() >= 0
    assert mbe.mbe_batchEnd() >= 0
    assert got.tobytes() == want.tobytes()
    assert np.concatenate([cur, prev, enh]).tobytes() == want_state.tobytes()
