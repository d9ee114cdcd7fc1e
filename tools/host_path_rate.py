#!/usr/bin/env python3
"""Development aid (GPU box): PCIe-inclusive rate of the host-buffer convenience call
mbx_process_batch_host (allocation + H2D + three launches + D2H, synchronous) -- never bench.py's value."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbelib_neo_amd import decoder, framegen  # noqa: E402
from mbelib_neo_amd.layout import init_state, rng_seeded  # noqa: E402

S, T = 65536, 1
frames = framegen.imbe_clean_voiced_frames(S * T, framegen.rng_for(7))
state, rng = init_state(S), rng_seeded(np.arange(S) + 1234)
out = decoder.process_batch_host(0, S, T, frames, state, rng)
state, rng = out["state"], out["rng"]
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out = decoder.process_batch_host(0, S, T, frames, state, rng)
    state, rng = out["state"], out["rng"]
dt = (time.perf_counter() - t0) / reps
print(f"mbx_process_batch_host: {S*T/dt/1e6:.2f} M frames/s ({dt*1e3:.1f} ms per call; moves {S*3*2604*2/1e6:.0f} MB of state, "
      f"{S*T*(18+320+640+20+16)/1e6:.0f} MB of frames/PCM/results over PCIe, plus allocation)")
