#!/usr/bin/env python3
"""time_decoder.py <codec> <S> <T> <resident 0|1> [min_ms=300] -- GPU box: ms per mbx_process_batch[_resident] launch of S streams x T
random-bit frames (steady state: ~30 ms of untimed launches, then >= min_ms between two events), and the kernel it takes.  For shapes
bench.py has no workload for (e.g. long launches on RESIDENT state: sessions / queue mode with several frames per channel and tick).
Development aid; MBX_HIP_LIBRARY selects a variant library."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mbelib_neo_amd import _native, decoder, framegen  # noqa: E402


def main():
    codec, S, T, resident = (int(x) for x in sys.argv[1:5])
    min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 300.0
    frames = framegen.random_frames(codec, S * T, framegen.rng_for(4242 + codec))
    dec = decoder.BatchDecoder(codec, S, seeds=np.arange(S) + 1234, resident=bool(resident))
    d_frames = dec.to_device(frames)
    out = dec.make_outputs(T, want_pcm16=True, want_float=False, want_results=True)
    L = _native.lib()
    for _ in range(3):
        dec.decode(d_frames, T, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dec.decode(d_frames, T, out=out)
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 5, 1e-6)
    for _ in range(max(1, int(0.03 / est))):
        dec.decode(d_frames, T, out=out)
    n = max(5, int(min_ms * 1e-3 / est))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        dec.decode(d_frames, T, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    name = (L.mbx_stream_kernel_name(codec, -T if resident else T) if T > 1 else L.mbx_batch_kernel_name(codec, S, T, resident)).decode()
    print(f"codec {codec} S {S} T {T} resident {resident}: {ms:.4f} ms per launch ({S * T / ms / 1e3:.1f} M frames/s), {n} launches, kernel {name}")


if __name__ == "__main__":
    main()
