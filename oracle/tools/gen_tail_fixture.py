#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, authoring container only (needs oracle/_ref: `make -C oracle ref fma`).
gpurun_out/tail_raw.npz (tools/find_tail.py on the GPU box: the frames of >= 31 M samples per codec whose int16 PCM
differs most between the HIP path and the oracle, with the inputs of their streams)  ->  tests/golden/tail_cases.npz:
per case the stream's wire frames up to the frame, its seed, and the frame's int16 PCM from
    ref_ieee   the REFERENCE, oracle/_ref/libmbe_ref.so      (-O2 -ffp-contract=off: the build the oracle restates)
    ref_fma    the REFERENCE, oracle/_ref/libmbe_ref_fma.so  (its own -std=gnu99 defaults for an FMA target, x86-64-v3)
    oracle     oracle/liboracle.so
    hip        what the HIP path produced when the case was found (tests/ recompute it)
Data only: inputs and outputs, no reference text."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402  (unpack_cells)
import oracle_lib  # noqa: E402

from mbelib_neo_amd.layout import FRAME_CELLS, init_state  # noqa: E402


def ref_run(libname, codec, frames, seed):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", libname))
    lib.ref_process_batch.restype = C.c_int
    lib.ref_process_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    T = frames.shape[0]
    cells = np.ascontiguousarray(bench.unpack_cells(codec, frames))
    ncell = FRAME_CELLS[codec][0] * FRAME_CELLS[codec][1]
    state = np.ascontiguousarray(init_state(1))
    pcm = np.zeros((T, 160), dtype=np.int16)
    rc = lib.ref_process_batch(codec, 1, T, cells.ctypes.data, ncell, state.ctypes.data, int(seed), pcm.ctypes.data)
    assert rc == 0, rc
    return pcm


def main():
    raw = np.load(os.path.join(ROOT, "gpurun_out", "tail_raw.npz"))
    o = oracle_lib.load()
    out = {"n": raw["n"]}
    for c in (0, 1, 2, 3):
        out[f"hist{c}"] = raw[f"hist{c}"]
    for k in range(int(raw["n"])):
        diff, codec, _, stream, t, seed = (int(x) for x in raw[f"c{k}_meta"])
        frames = raw[f"c{k}_frames"]
        ieee = ref_run("libref_bench.so", codec, frames, seed)[t]
        fma = ref_run("libref_bench_fma.so", codec, frames, seed)[t]
        ora = o.process_batch(codec, 1, t + 1, frames.reshape(t + 1, -1), o.init_state(1), o.rng_seeded([seed]))
        ora16 = np.asarray(ora["pcm16"]).reshape(t + 1, 160)[t]
        assert np.array_equal(ora16, raw[f"c{k}_ora16"]), "oracle does not reproduce the case"
        hip = raw[f"c{k}_hip16"]
        d = lambda a, b: int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max())   # noqa: E731
        print(f"case {k}: codec {codec} stream {stream} frame {t}: |hip-oracle| {d(hip, ora16)} |hip-ref_ieee| {d(hip, ieee)} "
              f"|oracle-ref_ieee| {d(ora16, ieee)} |ref_fma-ref_ieee| {d(fma, ieee)} peak|pcmf| {np.abs(raw[f'c{k}_oraf']).max():.0f}")
        out[f"c{k}_meta"] = np.array([codec, t, seed, diff], dtype=np.int64)
        out[f"c{k}_frames"] = frames
        out[f"c{k}_ref_ieee"] = ieee
        out[f"c{k}_ref_fma"] = fma
        out[f"c{k}_oracle"] = ora16
        out[f"c{k}_hip"] = hip
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tail_cases.npz"), **out)


if __name__ == "__main__":
    main()
