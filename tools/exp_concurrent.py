#!/usr/bin/env python3
"""experiment: does splitting the batch over several HIP streams (concurrent kernels) change throughput?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mbelib_neo_amd import _native, decoder, framegen

S = 65536
L = _native.lib()
for parts in (1, 2, 4, 8):
    Sp = S // parts
    decs, frames, outs, streams = [], [], [], []
    for p in range(parts):
        d = decoder.BatchDecoder(0, Sp, seeds=np.arange(Sp) + 1234)
        f = d.to_device(framegen.imbe_clean_voiced_frames(Sp, framegen.rng_for(5 + p)))
        decs.append(d); frames.append(f); outs.append(d.make_outputs(1)); streams.append(torch.cuda.Stream())
    L.mbx_reserve(S)
    def step():
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                st = streams[p].cuda_stream
                o = outs[p]
                # note: the expand workspace is shared, so expand + stream are issued per part in order on its stream;
                # parts race on the workspace -> results are garbage, timing only
                L.mbx_fec_imbe7200x4400(frames[p].data_ptr(), Sp, o["records"].data_ptr(), st)
                L.mbx_expand_records(0, o["records"].data_ptr(), Sp, st)
                L.mbx_stream_expanded(0, Sp, 1, o["records"].data_ptr(), decs[p].state.data_ptr(), decs[p].rng.data_ptr(),
                                      o["pcm16"].data_ptr(), None, o["results"].data_ptr(), st)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"parts={parts}: {dt*1e3:.3f} ms/step  {S/dt/1e6:.1f} Mframes/s")
