#!/usr/bin/env python3
"""Development aid (GPU box): one saved decision-flip stream (tools/cases/flip_*.npz, written by tools/find_flips.py)
frame by frame: where the device state first parts from the oracle's, bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import torch  # noqa: E402

from mbelib_neo_amd import decoder  # noqa: E402

d = np.load(sys.argv[1])
fr, seed, codec = d["frames"], int(d["seed"]), int(d["codec"])
o = oracle_lib.load()
st, rg = o.init_state(1), o.rng_seeded([seed])
dec = decoder.BatchDecoder(codec, 1, seeds=[seed])
for t in range(fr.shape[0]):
    r = o.process_batch(codec, 1, 1, fr[t], st, rg)
    st, rg = r["state"], r["rng"]
    dec.decode(np.ascontiguousarray(fr[t]), 1)
    torch.cuda.synchronize()
    g = dec.state_numpy()
    L = int(st[0, 1]["L"])
    line = [f"t {t} L {L} flags 0x{int(r['results']['flags'][0]):02x}"]
    for which, name in ((1, "prev(pre-enh)"), (0, "cur")):
        for field in ("Ml", "log2Ml"):
            a = st[0, which][field][1:L + 1].view(np.uint32).astype(np.int64)
            b = g[0, which][field][1:L + 1].view(np.uint32).astype(np.int64)
            line.append(f"{name}.{field} max ulp {int(np.abs(a - b).max())}")
        line.append(f"{name}.Vl equal {np.array_equal(st[0, which]['Vl'], g[0, which]['Vl'])}")
    line.append(f"localEnergy {st[0, 0]['localEnergy']!r} vs {g[0, 0]['localEnergy']!r}")
    print("  ".join(line))
