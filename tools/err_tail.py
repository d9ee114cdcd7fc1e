#!/usr/bin/env python3
"""Development aid (GPU box): the TAIL of the error distribution of the HIP path against the CPU oracle --
histogram of int16 differences and the largest float differences with their context, over many frames.
usage: tools/err_tail.py [codec] [S] [T] [seed]"""
import os
import sys

import numpy as np

ROOT = os.environ.get("MBX_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

from mbelib_neo_amd import decoder, framegen  # noqa: E402
from mbelib_neo_amd.layout import init_state, rng_seeded  # noqa: E402


def main():
    codec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 4242
    o = oracle_lib.load()
    rng = framegen.rng_for(seed)
    frames = framegen.random_frames(codec, S * T, rng)
    seeds = [99 + 7 * s for s in range(S)]
    ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(seeds))
    got = decoder.process_batch_host(codec, S, T, frames, init_state(S), rng_seeded(seeds))
    rf = np.asarray(ref["pcmf"], dtype=np.float64).reshape(-1, 160)
    gf = np.asarray(got["pcmf"], dtype=np.float64).reshape(-1, 160)
    d16 = np.abs(np.asarray(ref["pcm16"], dtype=np.int32).reshape(-1, 160) - np.asarray(got["pcm16"], dtype=np.int32).reshape(-1, 160))
    print(f"tree {ROOT} codec {codec} S={S} T={T}: int16 hist {np.bincount(d16.reshape(-1), minlength=6)[:8].tolist()}")
    e = np.abs(rf - gf)
    flags = np.asarray(ref["results"]["flags"]).reshape(-1)
    order = np.argsort(e.max(axis=1))[::-1][:6]
    for f in order:
        n = int(np.argmax(e[f]))
        st = ref["state"]
        print(f"  frame {f} (stream {f // T}, t {f % T}) sample {n}: err {e[f, n]:.4f} ref {rf[f, n]:.3f} frame peak {np.abs(rf[f]).max():.1f} "
              f"frame rms err {np.sqrt(np.mean(e[f] ** 2)):.4f} flags 0x{int(flags[f]):02x} int16 diff max {int(d16[f].max())}")


if __name__ == "__main__":
    main()
