#!/usr/bin/env python3
"""Development aid (GPU box): the (codec, kind) cases of tools/soak.py for one round, hard-decision kinds only, printing
every stream whose integer state (Vl) or PCM parts from the oracle -- candidates for float-threshold decision flips --
and saving the stream's frames + seed to gpurun_out/flip_<round>_<codec>_<kind>_<stream>.npz.
usage: tools/find_flips.py <first round> <rounds> [codecs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

from mbelib_neo_amd import decoder, framegen  # noqa: E402
from mbelib_neo_amd.layout import init_state, rng_seeded  # noqa: E402


def main():
    first, rounds = int(sys.argv[1]), int(sys.argv[2])
    codecs = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3]
    o = oracle_lib.load()
    total = 0
    for r in range(first, first + rounds):
        for codec in codecs:
            for kind in ("random", "clean"):
                S, T = 2048, 8
                rng = framegen.rng_for(90000 + 1000 * r + 10 * codec + len(kind))
                seeds = [77 + 13 * s + r for s in range(S)]
                frames = framegen.random_frames(codec, S * T, rng)
                if kind == "clean":
                    for _ in range(3):
                        frames &= framegen.random_frames(codec, S * T, rng)
                ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(seeds))
                got = decoder.process_batch_host(codec, S, T, frames, init_state(S), rng_seeded(seeds))
                rf = np.asarray(ref["pcmf"], dtype=np.float64).reshape(S, T, 160)
                gf = np.asarray(got["pcmf"], dtype=np.float64).reshape(S, T, 160)
                level = np.sqrt(np.mean(rf ** 2)) + 1e-30
                ratio = np.sqrt(np.mean((rf - gf) ** 2, axis=2)) / np.maximum(np.sqrt(np.mean(rf ** 2, axis=2)), 0.05 * level)
                bad = np.nonzero(ratio.max(axis=1) > 5e-3)[0]
                total += S * T
                for s in bad:
                    fb = frames.reshape(S, T, -1)[s]
                    print(f"round {r} codec {codec} {kind}: stream {s} frames {np.nonzero(ratio[s] > 5e-3)[0].tolist()} ratio {ratio[s].max():.3e}", flush=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", f"flip_{r}_{codec}_{kind}_{s}.npz"), frames=fb, seed=seeds[s], codec=codec)
        print("round", r, "done", total, "frames", flush=True)


if __name__ == "__main__":
    main()
