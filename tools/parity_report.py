#!/usr/bin/env python3
"""Development aid (GPU box): error statistics of the HIP path against the CPU oracle on seeded
random streams -- relative RMS, worst frame, histogram of int16 differences."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import parity  # noqa: E402

import mbelib_neo_amd as mbx  # noqa: E402
from mbelib_neo_amd import decoder, framegen  # noqa: E402
from mbelib_neo_amd.layout import init_state, rng_seeded  # noqa: E402


def main():
    o = oracle_lib.load()
    for codec, S, T, kind in [(0, 512, 16, "random"), (1, 512, 16, "random"), (0, 2048, 2, "voiced")]:
        rng = framegen.rng_for(1000 + 10 * codec + T)
        if kind == "voiced":
            f0 = framegen.imbe_clean_voiced_frames(S, rng)
            f1 = framegen.imbe_clean_voiced_frames(S, rng)
            frames = np.stack([f0, f1], axis=1).reshape(S * 2, 18)
        else:
            frames = framegen.random_frames(codec, S * T, rng)
        seeds = [1234 + s for s in range(S)]
        ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(seeds))
        got = decoder.process_batch_host(codec, S, T, frames, init_state(S), rng_seeded(seeds))
        rf = np.asarray(ref["pcmf"], dtype=np.float64).reshape(-1, 160)
        gf = np.asarray(got["pcmf"], dtype=np.float64).reshape(-1, 160)
        d = np.abs(np.asarray(ref["pcm16"], dtype=np.int32).reshape(-1) - np.asarray(got["pcm16"], dtype=np.int32).reshape(-1))
        level = np.sqrt(np.mean(rf ** 2))
        err = np.sqrt(np.mean((rf - gf) ** 2, axis=1))
        ratio = err / np.maximum(np.sqrt(np.mean(rf ** 2, axis=1)), 0.05 * level)
        hist = np.bincount(d, minlength=6)[:8]
        worst = int(np.argmax(np.abs(rf - gf).max(axis=1)))
        print(f"codec {codec} {kind} S={S} T={T}: rel_rms {parity.rel_rms(rf, gf):.3e} worst_frame {ratio.max():.3e} "
              f"max_abs_err {np.abs(rf - gf).max():.4f} (frame {worst}, frame peak {np.abs(rf[worst]).max():.1f}) int16 hist {hist.tolist()}")


if __name__ == "__main__":
    main()
