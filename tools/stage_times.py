#!/usr/bin/env python3
"""stage_times.py <workload> -- GPU box, DEVELOPMENT BUILD ONLY: where the waves of a one-frame launch spend their lives.
Build:  make -C mbelib-neo_amd/csrc OUT=$PWD/mbelib-neo_amd/libmbx_hip_stage.so EXTRA=-DMBX_STAGE_TIMES $PWD/mbelib-neo_amd/libmbx_hip_stage.so
Run:    MBX_HIP_LIBRARY=$PWD/mbelib-neo_amd/libmbx_hip_stage.so MBX_HIP_LIBRARY_ALLOW_OLDER=1 python tools/stage_times.py imbe_voiced
Every wave of the one-frame IMBE body writes the 100 MHz wall-clock time between consecutive marks (MBX_TS in mbx_stream.hip) to a
slot of its own (16 words per wave of the launch); this runs the bench workload for a few steps and prints mean / median per stage
over the waves of the LAST launch, and the distribution of wave lives.  The marks drain the scalar / LDS queues (s_memrealtime is
waited for) and cost a store each: the instrumented kernel is about 3 % slower than the product's."""
import ctypes as C
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {14: "(fused) entry -> the frame's bytes are there", 15: "(fused) -> C0 corrected, L known, table reads requested",
         10: "(fused) -> FEC done, record stored", 11: "(fused) -> table values there, expanded",
         1: "entry -> cur_mp's scalars there (first round trip)", 2: "-> frame parameters in LDS", 3: "-> decoded, policy applied",
         4: "-> snapshot stored, enhanced", 5: "-> smoothing, phases, bank coefficients", 6: "-> voiced bank (+ output through LDS)",
         7: "-> noise samples (table round trip)", 8: "-> transform pair (or nothing)", 9: "-> overlap-add",
         12: "-> soft clip, back in the body", 13: "-> every store issued"}
FRONT = {1: "entry -> the frame's words are there", 2: "-> C0 corrected (one table read)", 3: "-> demodulated (sequence window read)",
         4: "-> corrections there (one table read)", 5: "-> record assembled, b0 entry there: L known", 6: "-> L round requested, block info there, first cosine row there",
         7: "-> bit layout there, words scattered", 8: "-> B2 there, gains", 9: "-> inverse DCTs (a cosine row per output, one ahead)", 10: "-> rows written through (drained)"}
# (a mark a wave does not pass -- the transform pair of an all-voiced frame has none of its own -- keeps the value of an earlier launch
#  or zero; marks 7 / 8 are only meaningful for workloads with unvoiced bands)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "imbe_voiced"
    import bench
    from mbelib_neo_amd import _native
    lib = _native.lib()
    fn = lib.mbx_debug_stage_times
    fn.argtypes = [C.c_void_p, C.c_int]
    sys.argv = ["bench.py", "--workload", wl, "--steps", "6", "--warmup", "2", "--min-time-ms", "0", "--no-cpu-baseline", "--no-extras"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    import numpy as np
    S = int(line["config"]["streams_per_gpu"])
    marks = np.zeros((S, 16), dtype=np.uint32)
    assert fn(marks.ctypes.data, S) == 0
    t = marks.astype(np.float64) * 0.01
    if "one_launch" in line["roofline"]["kernel"]:
        # the grid is S/8 front blocks, then the stream blocks: the first rows of the buffer are FRONT blocks with marks of their own
        nf = (S + 7) // 8
        f = t[:nf]
        t = t[nf:]
        flife = f[:, sorted(FRONT)].sum(axis=1)
        print(f"{wl}: {nf} front blocks (eight frames each): life mean {flife.mean():.2f} us, median {np.median(flife):.2f}, "
              f"p10 {np.percentile(flife, 10):.2f}, p90 {np.percentile(flife, 90):.2f}")
        for i in sorted(FRONT):
            print(f"  {FRONT[i]:60s} mean {f[:, i].mean():6.2f} us  median {np.median(f[:, i]):6.2f}  {100.0 * f[:, i].mean() / flife.mean():5.1f} %")
        print(f"  ({len(t)} of the {S} stream blocks follow: the buffer holds {S} workgroups)")
    life = t[:, sorted(NAMES)].sum(axis=1)
    print(f"{wl}: kernel {line['roofline']['kernel']} {line['roofline']['kernel_ms']:.4f} ms (instrumented build), {len(t)} stream waves of the last launch")
    print(f"  wave life, entry to last store: mean {life.mean():.2f} us, median {np.median(life):.2f}, p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}")
    for i in NAMES:
        print(f"  {NAMES[i]:52s} mean {t[:, i].mean():6.2f} us  median {np.median(t[:, i]):6.2f}  {100.0 * t[:, i].mean() / life.mean():5.1f} %")


if __name__ == "__main__":
    main()
