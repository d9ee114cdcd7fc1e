#!/usr/bin/env python3
"""Development aid (GPU box): one case of tools/soak.py (round, codec, kind) in detail -- the worst frames and their context.
usage: tools/soak_case.py <round> <codec> <random|clean>"""
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
    r, codec, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    o = oracle_lib.load()
    S, T = 2048, 8
    rng = framegen.rng_for(90000 + 1000 * r + 10 * codec + len(kind))
    seeds = [77 + 13 * s + r for s in range(S)]
    frames = framegen.random_frames(codec, S * T, rng)
    if kind == "clean":
        for _ in range(3):
            frames &= framegen.random_frames(codec, S * T, rng)
    ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(seeds))
    got = decoder.process_batch_host(codec, S, T, frames, init_state(S), rng_seeded(seeds))
    rf = np.asarray(ref["pcmf"], dtype=np.float64).reshape(-1, 160)
    gf = np.asarray(got["pcmf"], dtype=np.float64).reshape(-1, 160)
    e = np.abs(rf - gf)
    flags = np.asarray(ref["results"]["flags"]).reshape(-1)
    print("records equal:", np.array_equal(got["records"]["w"], ref["records"]["w"]),
          " total rel rms %.3e" % (np.sqrt(np.mean((rf - gf) ** 2)) / np.sqrt(np.mean(rf ** 2))))
    order = np.argsort(e.max(axis=1))[::-1][:8]
    rs, gs = ref["state"].reshape(S, 3), got["state"].reshape(S, 3)
    for f in order:
        s, t = f // T, f % T
        n = int(np.argmax(e[f]))
        print(f"frame {f} (stream {s}, t {t}) sample {n}: err {e[f, n]:.4f} ref {rf[f, n]:.3f} got {gf[f, n]:.3f} frame peak {np.abs(rf[f]).max():.1f} "
              f"rms err {np.sqrt(np.mean(e[f] ** 2)):.4f} flags 0x{int(flags[f]):02x}")
    f = order[0]
    s = f // T
    print("stream", s, "per-frame max err:", [round(float(e[s * T + t].max()), 4) for t in range(T)])
    print("stream", s, "flags:", [hex(int(flags[s * T + t])) for t in range(T)])
    for name in ("w0", "L", "repeatCount", "errorRate", "noiseSeed", "localEnergy", "amplitudeThreshold"):
        print("  final", name, "ref", rs[s][0][name], "got", gs[s][0][name])
    for name in ("Ml", "PHIl", "PSIl", "log2Ml"):
        d = np.abs(rs[s][0][name].astype(np.float64) - gs[s][0][name].astype(np.float64))
        print("  final", name, "max abs diff %.3e at %d" % (d.max(), int(d.argmax())), " Vl equal:", np.array_equal(rs[s][0]["Vl"], gs[s][0]["Vl"]))


if __name__ == "__main__":
    main()
