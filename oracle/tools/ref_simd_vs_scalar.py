#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, authoring container only (needs oracle/_ref, i.e. /root/reference): run the REFERENCE's scalar
build and the REFERENCE's own SIMD build (`make -C oracle ref simd`) on the same seeded random-bit frames and count
where the two builds of the reference part: int16 differences, and frames that differ grossly (a float threshold
decision -- adaptive smoothing's Ml > VM -- taken the other way because the builds order their sums differently).
usage: oracle/tools/ref_simd_vs_scalar.py [simd|fma] [rounds] [codec ...]   (fma: the x86-64-v3 build of `make -C oracle fma`)"""
import ctypes as C
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(args):
    build, rnd, codec = args
    import bench   # unpack_cells
    from mbelib_neo_amd import framegen
    from mbelib_neo_amd.layout import FRAME_CELLS, init_state

    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", {"simd": "libref_bench_simd.so", "fma": "libref_bench_fma.so"}.get(build, "libref_bench.so")))
    lib.ref_process_batch.restype = C.c_int
    lib.ref_process_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    S, T = 2048, 8
    rng = framegen.rng_for(90000 + 1000 * rnd + 10 * codec + len("random"))   # the frames of tools/soak.py's "random" kind
    frames = framegen.random_frames(codec, S * T, rng)
    cells = np.ascontiguousarray(bench.unpack_cells(codec, frames))
    ncell = FRAME_CELLS[codec][0] * FRAME_CELLS[codec][1]
    state = np.ascontiguousarray(init_state(S))
    pcm = np.zeros((S * T, 160), dtype=np.int16)
    rc = lib.ref_process_batch(codec, S, T, cells.ctypes.data, ncell, state.ctypes.data, 77 + rnd, pcm.ctypes.data)
    assert rc == 0, rc
    return pcm


def main():
    other = "simd"
    if len(sys.argv) > 1 and sys.argv[1] in ("simd", "fma"):   # which second build: SIMD (`make -C oracle simd`) or FMA target (`make -C oracle fma`)
        other = sys.argv.pop(1)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    codecs = [int(c) for c in sys.argv[2:]] or [0, 1, 2, 3]
    tasks = [(b, r, c) for r in range(rounds) for c in codecs for b in ("scalar", other)]
    hist = {c: np.zeros(8, dtype=np.int64) for c in codecs}
    gross = {c: [] for c in codecs}
    frames = {c: 0 for c in codecs}
    with ProcessPoolExecutor(max_workers=8) as ex:
        out = list(ex.map(run, tasks))
    for i in range(0, len(tasks), 2):
        _, r, c = tasks[i]
        a, b = out[i].astype(np.int32), out[i + 1].astype(np.int32)
        d = np.abs(a - b)
        hist[c] += np.bincount(np.minimum(d.reshape(-1), 7), minlength=8)
        far = np.nonzero(d.max(axis=1) > 64)[0]
        gross[c] += [(r, int(f)) for f in far]
        frames[c] += a.shape[0]
    for c in codecs:
        print(f"codec {c}: {frames[c]} frames; int16 |scalar - {other}| histogram 0,1,..,6,>=7: {hist[c].tolist()}; "
              f"frames differing by more than 64: {len(gross[c])} {gross[c][:12]}")


if __name__ == "__main__":
    main()
