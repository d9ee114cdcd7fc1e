#!/usr/bin/env python3
"""find_tail.py (GPU box) -- the int16 tail of the HIP path against the CPU oracle over >= 10 M samples per codec, and the
INPUTS of the worst frames: for every frame whose int16 PCM differs from the oracle's by >= 3 LSB (and the three largest
per codec whatever they are) the stream's wire frames up to that frame, its seed and the HIP samples are written to
gpurun_out/tail_raw.npz.  oracle/tools/gen_tail_fixture.py (authoring container) turns that into
tests/golden/tail_cases.npz by running the REFERENCE's IEEE build and its FMA-target build on the same streams.
usage: tools/find_tail.py [S] [T] [seeds...]"""
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
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    seeds = [int(x) for x in sys.argv[3:]] or [4242]
    o = oracle_lib.load()
    cases = []
    summary = {}
    for codec in (0, 1, 2, 3):
        hist = np.zeros(16, dtype=np.int64)
        per_codec = []
        for seed in seeds:
            frames = framegen.random_frames(codec, S * T, framegen.rng_for(seed))
            sseeds = [99 + 7 * s for s in range(S)]
            ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(sseeds))
            got = decoder.process_batch_host(codec, S, T, frames, init_state(S), rng_seeded(sseeds))
            r16 = np.asarray(ref["pcm16"], dtype=np.int32).reshape(S, T, 160)
            g16 = np.asarray(got["pcm16"], dtype=np.int32).reshape(S, T, 160)
            gf = np.asarray(got["pcmf"], dtype=np.float32).reshape(S, T, 160)
            rf = np.asarray(ref["pcmf"], dtype=np.float32).reshape(S, T, 160)
            d = np.abs(r16 - g16)
            hist += np.bincount(np.minimum(d.reshape(-1), 15), minlength=16)
            worst = d.max(axis=2)
            fb = frames.reshape(S, T, -1)
            for s, t in np.argwhere(worst >= 2):
                per_codec.append((int(worst[s, t]), codec, seed, int(s), int(t), sseeds[s], fb[s, : t + 1].copy(), g16[s, t].astype(np.int16),
                                  gf[s, t].copy(), r16[s, t].astype(np.int16), rf[s, t].copy()))
        per_codec.sort(key=lambda c: -c[0])
        keep = [c for c in per_codec if c[0] >= 3] + [c for c in per_codec if c[0] < 3][:3]
        cases += keep[:24]
        summary[codec] = hist.tolist()
        print(f"codec {codec}: {hist.sum()} samples, int16 |hip - oracle| histogram {hist[:8].tolist()}, frames kept {len(keep[:24])}", flush=True)
    out = {"n": np.array(len(cases)), "S": np.array(S), "T": np.array(T)}
    for k, c in enumerate(cases):
        out[f"c{k}_meta"] = np.array([c[0], c[1], c[2], c[3], c[4], c[5]], dtype=np.int64)   # diff, codec, batch seed, stream, frame, stream seed
        out[f"c{k}_frames"] = c[6]
        out[f"c{k}_hip16"] = c[7]
        out[f"c{k}_hipf"] = c[8]
        out[f"c{k}_ora16"] = c[9]
        out[f"c{k}_oraf"] = c[10]
    for codec, h in summary.items():
        out[f"hist{codec}"] = np.array(h)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "tail_raw.npz"), **out)


if __name__ == "__main__":
    main()
