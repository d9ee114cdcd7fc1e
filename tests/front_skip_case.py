"""Child process of test_one_launch_fall_back_path_gives_the_same_bytes (never imported by pytest): runs with
MBX_HIP_LIBRARY=<...>/libmbx_hip_testing.so, the -DMBX_TESTING build of the product's sources, which alone exports the
fault-injection hook mbx_testing_set_front_skip.  For one codec: four ticks of 4,096 + 5 streams, ABI and resident state, once
undisturbed and once with every fourth front block of the one-launch kernel doing nothing -- both must give the same bytes, the
fall-back counter must have counted exactly the streams of the skipped chunks.  Prints one JSON line with the SHA-256 of the
undisturbed run's bytes per state form, which the parent compares with what the PRODUCT library gives on the same input."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def case_inputs(codec):
    from mbelib_neo_amd import framegen
    from mbelib_neo_amd.layout import FRAME_BYTES

    fb = FRAME_BYTES[codec]
    S, T = 4096 + 5, 4
    frames = framegen.random_frames(codec, S * T, framegen.rng_for(0x5A + codec)).reshape(S, T, fb)
    frames[::3] &= framegen.random_frames(codec, ((S + 2) // 3) * T, framegen.rng_for(0x5B + codec)).reshape(-1, T, fb)
    return S, T, frames, np.arange(S) + 5


def run(codec, resident, before_launches=None):
    """four ticks through mbx_process_batch[_resident]; returns (bytes digest, parts) of everything the launches wrote"""
    from mbelib_neo_amd import decoder

    S, T, frames, seeds = case_inputs(codec)
    if before_launches:
        before_launches()
    dec = decoder.BatchDecoder(codec, S, seeds=seeds, resident=resident)
    h = hashlib.sha256()
    for t in range(T):
        o = dec.decode(np.ascontiguousarray(frames[:, t]), 1, want_float=True)
        for k in ("records", "results", "pcm16", "pcmf"):
            h.update(o[k].cpu().numpy().tobytes())
    h.update(dec.state_numpy().tobytes())
    h.update(dec.rng_numpy().tobytes())
    return h.hexdigest()


def main():
    import torch

    import mbelib_neo_amd as m
    from mbelib_neo_amd import decoder

    codec = int(sys.argv[1])
    L = m.lib()
    assert hasattr(L, "mbx_testing_set_front_skip"), "not the testing build: " + str(os.environ.get("MBX_HIP_LIBRARY"))
    decoder.ensure_init(0)
    strm = torch.cuda.current_stream().cuda_stream
    S, T, _, _ = case_inputs(codec)
    assert L.mbx_testing_set_front_skip(3) == -1   # (a power of two, or 0)
    out = {}
    for resident in (False, True):
        ref = run(codec, resident)
        before = L.mbx_front_fallbacks(strm)
        try:
            got = run(codec, resident, lambda: L.mbx_testing_set_front_skip(4))
        finally:
            assert L.mbx_testing_set_front_skip(0) == 0
        after = L.mbx_front_fallbacks(strm)
        chunks = (S + 7) // 8
        skipped_streams = sum(min(8, S - 8 * c) for c in range(0, chunks, 4))
        assert got == ref, (resident, "the fall-back path gave other bytes")
        assert after - max(before, 0) == T * skipped_streams, (before, after, skipped_streams)
        out["resident" if resident else "abi"] = ref
    out["fallbacks_counted"] = int(after)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
