"""Readers for the golden fixture files in tests/golden (written by oracle/tools/gen_fixtures.c
linked against the real reference; layouts documented there and in tests/golden/README.md)."""
import os

import numpy as np

from mbelib_neo_amd.layout import PARMS_DTYPE, RESULT_DTYPE

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read(name):
    return np.fromfile(os.path.join(HERE, name), dtype=np.uint8)


def ecc_kat():
    b = _read("ecc_kat.bin")
    rec = np.dtype([("inp", "<u4"), ("out", "<u4"), ("errs", "<i4")])
    ng = int(b[:4].view("<u4")[0])
    golay = b[4 : 4 + ng * 12].view(rec)
    off = 4 + ng * 12
    nh = int(b[off : off + 4].view("<u4")[0])
    ham = b[off + 4 : off + 4 + nh * 12].view(rec)
    return golay, ham


def fec(codec):
    b = _read("fec_imbe.bin" if codec == 0 else "fec_ambe.bin")
    ncell, nd = (184, 88) if codec == 0 else (96, 49)
    rec = np.dtype([("cells", "i1", (ncell,)), ("bits", "i1", (nd,)), ("ret", "<i4"), ("result", RESULT_DTYPE)])
    n = int(b[:4].view("<u4")[0])
    return b[4 : 4 + n * rec.itemsize].view(rec)


def stream(codec):
    b = _read("stream_imbe.bin" if codec == 0 else "stream_ambe.bin")
    ncell, nd = (184, 88) if codec == 0 else (96, 49)
    S, T = (int(x) for x in b[:8].view("<u4"))
    frame = np.dtype(
        [
            ("cells", "i1", (ncell,)),
            ("bits", "i1", (nd,)),
            ("ret", "<i4"),
            ("result", RESULT_DTYPE),
            ("pcmf", "<f4", (160,)),
            ("pcm16", "<i2", (160,)),
            ("digest", "<u4", (3,)),
        ]
    )
    per_stream = np.dtype([("frames", frame, (T,)), ("final", PARMS_DTYPE, (3,))])
    return S, T, b[8 : 8 + S * per_stream.itemsize].view(per_stream)


def golden_synth():
    b = _read("golden_synth.bin")
    d = np.dtype(
        [
            ("cur_in", PARMS_DTYPE),
            ("prev_in", PARMS_DTYPE),
            ("pcmf", "<f4", (160,)),
            ("pcm16", "<i2", (160,)),
            ("hash_f32", "<u4"),
            ("hash_s16", "<u4"),
            ("cur_out", PARMS_DTYPE),
            ("prev_out", PARMS_DTYPE),
        ]
    )
    return b.view(d)[0]


def synth_seq():
    b = _read("synth_seq.bin")
    nrec, frames = (int(x) for x in b[:8].view("<u4"))
    d = np.dtype([("pcmf", "<f4", (frames, 160)), ("cur", PARMS_DTYPE), ("prev", PARMS_DTYPE)])
    return frames, b[8:].view(d)


def f2s():
    b = _read("f2s_kat.bin")
    n = int(b[:4].view("<u4")[0])
    d = np.dtype([("inp", "<f4", (160,)), ("out", "<i2", (160,))])
    return b[4 : 4 + n * d.itemsize].view(d)


def params_kat():
    b = _read("params_kat.bin")
    imbe = np.dtype([("rc", "<i4"), ("w0", "<f4"), ("L", "<i4"), ("K", "<i4")])
    ambe = np.dtype([("rc", "<i4"), ("w0", "<f4"), ("L", "<i4")])
    off = 0
    t_imbe = b[off : off + 256 * imbe.itemsize].view(imbe)
    off += 256 * imbe.itemsize
    t_ambe = b[off : off + 128 * ambe.itemsize].view(ambe)
    off += 128 * ambe.itemsize
    full = []
    for nd in (88, 49):
        d = np.dtype([("bits", "i1", (nd,)), ("prev_L", "<i4"), ("rc", "<i4"), ("cur", PARMS_DTYPE)])
        full.append(b[off : off + 64 * d.itemsize].view(d))
        off += 64 * d.itemsize
    assert off == len(b)
    return t_imbe, t_ambe, full[0], full[1]


def misc_kat():
    b = _read("misc_kat.bin")
    d = np.dtype(
        [
            ("tm", "<i4"),
            ("ml1", "<f4"),
            ("local_energy", "<f4"),
            ("comfort", "<f4", (160,)),
            ("cold_seed", "<f4"),
            ("hr_rc", "<i4"),
            ("hr_result", RESULT_DTYPE),
            ("hr_cur", PARMS_DTYPE),
            ("hr_pcm", "<f4", (160,)),
        ]
    )
    assert d.itemsize == len(b)
    return b.view(d)[0]


def soft_kat():
    """soft_kat.bin (layout: oracle/tools/gen_fixtures.c gen_soft): dict of structured arrays"""
    b = _read("soft_kat.bin")
    out, off = {}, 0

    def section(name, dt):
        nonlocal off
        n = int(b[off : off + 4].view("<u4")[0])
        off += 4
        out[name] = b[off : off + n * dt.itemsize].view(dt)
        off += n * dt.itemsize

    section("golay", np.dtype([("soft", "u1", (23, 2)), ("out", "i1", (23,)), ("ret", "<i4")]))
    section("hamming", np.dtype([("soft", "u1", (15, 2)), ("out", "i1", (15,)), ("ret", "<i4")]))
    section("imbe", np.dtype([("soft", "u1", (184, 2)), ("bits", "i1", (88,)), ("ret", "<i4"), ("result", RESULT_DTYPE)]))
    section("ambe", np.dtype([("soft", "u1", (96, 2)), ("bits", "i1", (49,)), ("ret", "<i4"), ("result", RESULT_DTYPE)]))
    section("llr", np.dtype([("llr", "<i2"), ("soft", "u1", (2,))]))
    section("process", np.dtype([("soft", "u1", (184, 2)), ("ret", "<i4"), ("result", RESULT_DTYPE), ("pcmf", "<f4", (160,))]))
    assert off == b.size
    return out


def imbe7100_kat():
    """imbe7100_kat.bin (layout: oracle/tools/gen_fixtures.c gen_imbe7100)"""
    b = _read("imbe7100_kat.bin")
    out, off = {}, 0

    def section(name, dt):
        nonlocal off
        n = int(b[off : off + 4].view("<u4")[0])
        off += 4
        out[name] = b[off : off + n * dt.itemsize].view(dt)
        off += n * dt.itemsize

    section("hamming", np.dtype([("inp", "<u4"), ("out", "<u4"), ("errs", "<i4")]))
    section("convert", np.dtype([("inp", "i1", (88,)), ("out", "i1", (88,))]))
    section("fec", np.dtype([("cells", "i1", (168,)), ("bits", "i1", (88,)), ("ret", "<i4"), ("result", RESULT_DTYPE)]))
    S, T = (int(x) for x in b[off : off + 8].view("<u4"))
    off += 8
    frame = np.dtype([("cells", "i1", (168,)), ("ret", "<i4"), ("result", RESULT_DTYPE), ("pcmf", "<f4", (160,))])
    per_stream = np.dtype([("frames", frame, (T,)), ("final", PARMS_DTYPE)])
    out["stream"] = b[off : off + S * per_stream.itemsize].view(per_stream)
    off += S * per_stream.itemsize
    section("hamming_soft", np.dtype([("soft", "u1", (15, 2)), ("out", "i1", (15,)), ("ret", "<i4")]))
    section("fec_soft", np.dtype([("soft", "u1", (168, 2)), ("bits", "i1", (88,)), ("ret", "<i4"), ("result", RESULT_DTYPE)]))
    assert off == b.size
    return out


def ambe2400_kat():
    """ambe2400_kat.bin (layout: oracle/tools/gen_fixtures.c gen_ambe2400): (frame-level streams, data-level streams)"""
    b = _read("ambe2400_kat.bin")
    off = 0
    S, T = (int(x) for x in b[off : off + 8].view("<u4"))
    off += 8
    frame = np.dtype([("cells", "i1", (96,)), ("bits", "i1", (49,)), ("ret", "<i4"), ("result", RESULT_DTYPE), ("pcmf", "<f4", (160,))])
    per_stream = np.dtype([("frames", frame, (T,)), ("final", PARMS_DTYPE, (3,))])
    framed = b[off : off + S * per_stream.itemsize].view(per_stream)
    off += S * per_stream.itemsize
    S2, T2 = (int(x) for x in b[off : off + 8].view("<u4"))
    off += 8
    dframe = np.dtype([("bits", "i1", (49,)), ("total_in", "<i4"), ("ret", "<i4"), ("result", RESULT_DTYPE), ("pcmf", "<f4", (160,))])
    per_stream2 = np.dtype([("frames", dframe, (T2,)), ("final", PARMS_DTYPE, (3,))])
    data = b[off : off + S2 * per_stream2.itemsize].view(per_stream2)
    off += S2 * per_stream2.itemsize
    assert off == b.size
    return framed, data


def tone_kat():
    """tone_kat.bin (layout: oracle/tools/gen_fixtures.c gen_tones): (AMBE+2 tone frames, D-STAR tone frames)"""
    b = _read("tone_kat.bin")
    d1 = np.dtype([("bits", "i1", (49,)), ("pcmf", "<f4", (160,)), ("swn", "<i4"), ("tonePhase", "<u4")])
    d2 = np.dtype([("id", "<i4"), ("pcmf", "<f4", (160,)), ("swn", "<i4"), ("tonePhase", "<u4")])
    n1 = int(b[:4].view("<u4")[0])
    a = b[4 : 4 + n1 * d1.itemsize].view(d1)
    off = 4 + n1 * d1.itemsize
    n2 = int(b[off : off + 4].view("<u4")[0])
    c = b[off + 4 : off + 4 + n2 * d2.itemsize].view(d2)
    assert off + 4 + n2 * d2.itemsize == b.size
    return a, c


def notones_kat():
    """notones_kat.bin (layout: oracle/tools/gen_fixtures.c gen_notones; written by the reference built with its NOTONES option):
    (AMBE+2 tone frames, D-STAR tone frames) as tone_kat(), then scripted data-level streams (AMBE+2 3600x2450, AMBE 3600x2400)"""
    b = _read("notones_kat.bin")
    d1 = np.dtype([("bits", "i1", (49,)), ("pcmf", "<f4", (160,)), ("swn", "<i4"), ("tonePhase", "<u4")])
    d2 = np.dtype([("id", "<i4"), ("pcmf", "<f4", (160,)), ("swn", "<i4"), ("tonePhase", "<u4")])
    n1 = int(b[:4].view("<u4")[0])
    a = b[4 : 4 + n1 * d1.itemsize].view(d1)
    off = 4 + n1 * d1.itemsize
    n2 = int(b[off : off + 4].view("<u4")[0])
    c = b[off + 4 : off + 4 + n2 * d2.itemsize].view(d2)
    off += 4 + n2 * d2.itemsize
    streams = []
    for _ in range(2):
        S, T = (int(x) for x in b[off : off + 8].view("<u4"))
        off += 8
        dframe = np.dtype([("bits", "i1", (49,)), ("total_in", "<i4"), ("ret", "<i4"), ("result", RESULT_DTYPE), ("pcmf", "<f4", (160,))])
        per_stream = np.dtype([("frames", dframe, (T,)), ("final", PARMS_DTYPE, (3,))])
        streams.append(b[off : off + S * per_stream.itemsize].view(per_stream))
        off += S * per_stream.itemsize
    assert off == b.size
    return a, c, streams[0], streams[1]
