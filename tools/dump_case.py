#!/usr/bin/env python3
"""Development aid (GPU box): dump oracle and HIP float PCM of one stream of a tools/soak.py case to gpurun_out/."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from mbelib_neo_amd import decoder, framegen
from mbelib_neo_amd.layout import init_state, rng_seeded
r, codec, kind, s = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
o = oracle_lib.load()
S, T = 2048, 8
rng = framegen.rng_for(90000 + 1000 * r + 10 * codec + len(kind))
frames = framegen.random_frames(codec, S * T, rng).reshape(S, T, -1)
fr = np.ascontiguousarray(frames[s]).reshape(T, -1)
seeds = [77 + 13 * s + r]
ref = o.process_batch(codec, 1, T, fr, o.init_state(1), o.rng_seeded(seeds))
got = decoder.process_batch_host(codec, 1, T, fr, init_state(1), rng_seeded(seeds))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "case_ref.npy"), np.asarray(ref["pcmf"]).reshape(T, 160))
np.save(os.path.join(ROOT, "gpurun_out", "case_got.npy"), np.asarray(got["pcmf"]).reshape(T, 160))
print("max err per frame", np.abs(np.asarray(ref["pcmf"]).reshape(T,160) - np.asarray(got["pcmf"]).reshape(T,160)).max(axis=1))
