#!/usr/bin/env python3
"""frame_stamps.py [calls=400] -- GPU box, DEVELOPMENT BUILD ONLY (make -C mbelib-neo_amd/csrc EXTRA=-DMBX_FRAME_STAMPS):
where a synchronous mbe_processImbe7200x4400Framef call spends its time.  The single-frame body writes 100 MHz wall-clock
stamps (mbx_stream.hip, MBX_STAMP) which this reads back after every call (mbx_debug_frame_stamps); the host's own clock
brackets the call.  Prints medians, in microseconds, of the device-side stages and of the host-visible total.
Run once plain and once with MBE_NEO_FRAME_SERVER=1."""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import shim_lib  # noqa: E402
from shim_lib import p  # noqa: E402
from mbelib_neo_amd.layout import PARMS_DTYPE, RESULT_DTYPE  # noqa: E402

STAGES = ["entry -> FEC done (state loads in flight)", "-> state arrived, parked", "-> frame expanded", "-> decoded + policy",
          "-> enhanced", "-> synthesised", "-> all stores issued", "-> stores drained", "-> completion word out"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    mbe = shim_lib.load()
    hip = C.CDLL(os.path.join(ROOT, "mbelib-neo_amd", "libmbx_hip.so"))
    hip.mbx_debug_frame_stamps.argtypes = [C.c_void_p]
    st = np.zeros(3, dtype=PARMS_DTYPE)
    mbe.mbe_initMbeParms(p(st[0:1]), p(st[1:2]), p(st[2:3]))
    rng = np.random.default_rng(5)
    out = np.zeros(160, dtype=np.float32)
    d = np.zeros(88, dtype=np.int8)
    r = np.zeros(1, dtype=RESULT_DTYPE)
    stamps = np.zeros(16, dtype=np.uint64)
    rows, host = [], []
    # clean voiced-ish frames: random information bits through a frame with no channel errors is what host_bench times too
    for i in range(n + 20):
        fr = rng.integers(0, 2, size=(8, 23), dtype=np.int8)
        t0 = time.perf_counter()
        mbe.mbe_processImbe7200x4400Framef(p(out), p(r), p(fr), p(d), p(st[0:1]), p(st[1:2]), p(st[2:3]))
        t1 = time.perf_counter()
        if i % 4 == 3 and i >= 20:   # three undisturbed calls, then one that is read back (the read-back drains the device)
            assert hip.mbx_debug_frame_stamps(p(stamps)) == 0
            rows.append(stamps.astype(np.int64).copy())
            host.append((t1 - t0) * 1e6)
    rows = np.array(rows)
    print(f"calls read back: {len(rows)}; host-side call (python + ctypes around it) median {statistics.median(host):.2f} us")
    tot = (rows[:, 9] - rows[:, 0]) * 0.01
    print(f"device: entry -> completion word  median {np.median(tot):.2f} us  (min {tot.min():.2f})")
    for k, name in enumerate(STAGES):
        dt = (rows[:, k + 1] - rows[:, k]) * 0.01
        print(f"  {name:45s} {np.median(dt):6.2f} us")
    if rows[:, 10].any():   # inside the IMBE 7200x4400 FEC (stamps 10..13 of mbx_fec_frame.h)
        for a, b, name in ((0, 10, "entry -> wire frame in registers"), (10, 11, "-> c0 Golay"), (11, 12, "-> demodulation sequence"),
                           (12, 13, "-> c1..c6 decoded, record packed"), (13, 1, "-> record stored and broadcast")):
            print(f"    FEC: {name:41s} {np.median((rows[:, b] - rows[:, a]) * 0.01):6.2f} us")


if __name__ == "__main__":
    main()
