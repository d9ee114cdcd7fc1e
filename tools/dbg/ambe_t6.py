import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from mbelib_neo_amd import framegen, decoder
from mbelib_neo_amd.layout import init_state, rng_seeded
codec, S, T = 1, 1024, 6
frames = framegen.random_frames(codec, S * T, framegen.rng_for(1000 + 10 * codec + T))
seeds = [1234 + s for s in range(S)]
o = oracle_lib.load()
ref = o.process_batch(codec, S, T, frames, o.init_state(S), o.rng_seeded(seeds))
dec = decoder.BatchDecoder(codec, S, seeds=np.array(seeds))
out = dec.decode(frames, T, want_float=True)
got = out["pcmf"].cpu().numpy().reshape(S, T, 160)
rf = np.asarray(ref["pcmf"]).reshape(S, T, 160)
err = np.sqrt(((got - rf) ** 2).mean(axis=2))
lvl = np.sqrt((rf ** 2).mean(axis=2))
fl = np.asarray(ref["results"]["flags"]).reshape(S, T)
bad = np.argwhere(err > 1e-3 * (lvl + 1.0))
print("bad frames:", len(bad))
for s, t in bad[:20]:
    print("stream", s, "frame", t, "err", err[s, t], "lvl", lvl[s, t], "flags of stream", [hex(x) for x in fl[s]])
