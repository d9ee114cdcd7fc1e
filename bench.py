#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched MBE decode path on MI355X.

Metric (BASELINE.json): 20 ms frames/s, whole job, IMBE 7200x4400.
Workload (default, BASELINE.json configs[1]): 65,536 streams per GPU x T=1 frame per step,
clean-encoded all-voiced IMBE frames, state warmed by one identical frame so both the
previous and the current model are voiced.  One "step" = one pass of the hot path
(mbx_process_batch: for this shape ONE launch, imbe_one_launch_kernel -- front blocks do FEC and parameter expansion for
eight frames each, stream blocks the stream stage; other shapes: FEC kernel, [parameter-expansion kernel,] stream kernel) over the whole batch
with every input already resident in HBM.  After the timed region a strided sample of the timed streams is replayed through
the CPU oracle and the line carries the PCM error (`parity`: the metric's second half).  Streams are independent, so N GPUs = N independent
shards (weak scaling, no data-path collective); the only collective is the RCCL broadcast of the
constant-table blob at start-up (plus the timing reduction and the checksum all-gather).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) bench.py starts the N ranks itself
(child processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything touches a GPU) and
relays rank 0's line; under torch.distributed.run it is one of the ranks.  `--workload ambe_stream
--gpus 8` is BASELINE configs[4] (8 x 8,192 AMBE+2 streams x T = 128).

Prints ONE COMPACT JSON line on rank 0 (the contract keys, `roofline`, `parity`, `cpu_baseline`, and per other config
{value, ms_per_step, kernel, kernel_ms, frac}: at most 4 KB -- contract_line()) and writes everything else to the
sidecar `bench_detail.json` next to this script (and to gpurun_out/ when that directory exists): `other_configs` in full
(the other GPU configs of BASELINE.json, 10 steps each, with their SQ issue models), `cpu_baselines` (the reference's
scalar and SIMD builds, full path and its own bench_synth / bench_unvoiced recipes, one core and all cores), `host_path`
(what a C host sees from host memory to host memory), `infinity_cache_assisted` (the same workload in the library's
default alternating stream order; the headline itself is timed in a FIXED order, i.e. against HBM), `valu`, `copy_floor`.
Nothing but the compact line goes to stdout.  The timed region is at least --steps steps and at
least --min-time-ms (500 ms) of wall time: `steps_effective`; the warm-up ends with ~30 ms of untimed steps directly in front of it
(`settle_steps`: past the load-onset transient of the power controller, profiles/r05/launch_series.txt).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector fp32 (non-MFMA) peak

WORKLOADS = {
    # name: (codec, streams per GPU, T, description)
    "imbe_voiced": (0, 65536, 1, "BASELINE configs[1]: 65,536 IMBE 7200x4400 streams x T=1, clean all-voiced frames, warm state"),
    # the same workload on RESIDENT state (mbx_process_batch_resident: what sessions and the queue mode's pool use) -- reported
    # beside the headline with its own algorithmic bytes, never as it: the drop-in batch call keeps the ABI triplets whole
    "imbe_voiced_resident": (0, 65536, 1, "65,536 IMBE 7200x4400 streams x T=1, clean all-voiced frames, warm RESIDENT state (prev_mp_enhanced elided, prev_mp lazy)"),
    "imbe_mixed": (0, 65536, 16, "BASELINE configs[3]: 65,536 IMBE streams x T=16 random-bit frames (mixed voiced/unvoiced)"),
    "ambe_fec": (1, 65536, 1, "BASELINE configs[2]: 65,536 AMBE+2 streams x T=1, clean voice frames + 1% bit flips"),
    "ambe_fec_resident": (1, 65536, 1, "65,536 AMBE+2 streams x T=1, clean voice frames + 1% bit flips, RESIDENT state (prev_mp_enhanced elided, prev_mp lazy)"),
    "ambe_stream": (1, 8192, 128, "BASELINE configs[4] per-GPU shard: 8,192 AMBE+2 streams x T=128 random-bit frames, int16 out"),
    # SURVEY.md §8(f) row 4 (not BASELINE configs): the other two codecs of the reference
    "imbe7100_mixed": (2, 65536, 16, "65,536 IMBE 7100x4400 streams x T=16 random-bit frames"),
    "ambe2400_mixed": (3, 65536, 16, "65,536 AMBE 3600x2400 (D-STAR) streams x T=16 random-bit frames"),
    # SURVEY.md §8(f) row 1 (not a BASELINE config): the soft-decision front end in front of the same path
    "imbe_soft": (0, 65536, 1, "65,536 IMBE 7200x4400 streams x T=1, soft-decision frames (noisy observations of random bits)"),
    "ambe_soft": (1, 65536, 1, "65,536 AMBE+2 3600x2450 streams x T=1, soft-decision frames (noisy observations of random bits)"),
    # the same front end on what a receiver sees: encoded voice frames through noise (about 2.3 % wrong hard decisions)
    "imbe_soft_coded": (0, 65536, 1, "65,536 IMBE 7200x4400 streams x T=1, soft-decision frames (noisy observations of ENCODED voice frames)"),
    "ambe_soft_coded": (1, 65536, 1, "65,536 AMBE+2 3600x2450 streams x T=1, soft-decision frames (noisy observations of ENCODED voice frames)"),
}
CODEC_NAME = {0: "IMBE 7200x4400", 1: "AMBE+2 3600x2450", 2: "IMBE 7100x4400", 3: "AMBE 3600x2400"}


def make_frames(name, codec, S, T, rank):
    from mbelib_neo_amd import framegen

    rng = framegen.rng_for(0xBE0000 + 97 * rank + codec)
    if name in ("imbe_voiced", "imbe_voiced_resident"):
        return framegen.imbe_clean_voiced_frames(S * T, rng)
    if name in ("ambe_fec", "ambe_fec_resident"):
        return framegen.ambe_noisy_voice_frames(S * T, rng, ber=0.01)
    if name.endswith("_soft_coded"):
        return framegen.soft_frames_coded(codec, S * T, rng)
    if name.endswith("_soft"):
        return framegen.soft_frames(codec, S * T, rng)
    return framegen.random_frames(codec, S * T, rng)


def nominal_flops_per_frame(workload):
    """SURVEY.md §8(d)'s flop model (a count of the reference's arithmetic, not of this kernel's instructions):
    voiced bank 9 flop x 160 samples x voiced components, unvoiced path 12 kflop, everything else 4 kflop.
    imbe_voiced: prev and cur components voiced on all L harmonics, mean L over uniform b0 in [0, 207] = 32.35,
    no unvoiced band; every other workload: the survey's measured random-bit mix (36 kflop of voiced bank)."""
    if workload == "imbe_voiced":
        return 9 * 160 * 2 * 32.35 + 4000, "9*160*(2*mean L = 64.7 voiced components) + 4k, no unvoiced band"
    return 36000 + 12000 + 4000, "survey's random-bit mix: 36k voiced bank + 12k unvoiced FFT path + 4k"


def algorithmic_bytes_per_launch(codec, S, T, resident=False):
    """B_io = packed channel bits in + int16 PCM out per frame; B_state = load +
    store of the three-struct state per stream per launch.
    resident: what the resident form has to move per stream and launch -- cur_mp in and out (2 x 2604), prev_mp out
    (2604: the snapshot), and of prev_mp in only what the decode reads (Ml, log2Ml, PHIl: 3 x 57 floats, + 5 scalars =
    704 B); prev_mp_enhanced not at all."""
    b_io = (18 if codec in (0, 2) else 9) + 320
    if resident:
        return S * T * b_io + S * (3 * 2604 + 704)
    return S * T * b_io + S * 2 * 3 * 2604


def soft_fec_bytes_per_launch(codec, n):
    """soft-decision FEC kernel: the mbe_soft_bit frame in (2 B per cell) + the 16-byte parameter record out"""
    return n * ({0: 184, 1: 96, 2: 168, 3: 96}[codec] * 2 + 16)


def library_sha():
    from mbelib_neo_amd import _native

    return hashlib.sha256(open(_native.library_path(), "rb").read()).hexdigest()[:16]


def measured_traffic(workload, S, T):
    """HBM bytes per STEP (= per mbx_process_batch call) of the dominant kernel from a committed PMC summary of this workload and size
    (profiles/rNN/<workload>_pmc.json, written by tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 passes, each calibrated on state_copy_kernel's known byte count).  A step that is ONE dispatch of the kernel: that
    dispatch's bytes; a SLICED step (k dispatches of the same kernel on three queues): the sum over its k dispatches
    (`traffic_bytes_per_launch` = mean per dispatch x `dispatches_per_step`; round 5 reported one slice's bytes against a whole step's
    time).  PMC counters cannot be read from inside this process, so the summary must come from the SAME build: it records the sha256
    of libmbx_hip.so it was taken with, and a summary of another build gives (None, reason)."""
    import glob

    sha = library_sha()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("streams_per_gpu") == S and d.get("frames_per_stream_per_step") == T and "dominant_kernel" in d:
            rel = os.path.relpath(path, ROOT)
            if d.get("libmbx_hip_sha256_16") == sha:
                return d["dominant_kernel"]["traffic_bytes_per_launch"], rel
            stale = stale or f"{rel} was taken with another build of libmbx_hip.so ({d.get('libmbx_hip_sha256_16', 'unrecorded')} != {sha})"
    return None, stale or "no PMC summary for this workload and size"


def measured_issue(workload, S, T):
    """SQ-counter summary of the dominant kernel (profiles/rNN/<workload>_sq.json, written by tools/sq_profile.py on the GPU
    box: instruction counts by class, busy / wait quad-cycles, clock) -- like the traffic figure only when it was taken
    with the SAME build of libmbx_hip.so."""
    import glob

    sha = library_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_sq.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("streams_per_gpu") == S and d.get("frames_per_stream_per_step") == T and "issue_model" in d:
            if d.get("libmbx_hip_sha256_16") == sha:
                return dict(d["issue_model"], source=os.path.relpath(path, ROOT))
            return {"source": os.path.relpath(path, ROOT), "stale": f"taken with another build of libmbx_hip.so ({d.get('libmbx_hip_sha256_16')} != {sha})"}
    return None


def unpack_cells(codec, frames):
    """packed wire frames [n, 18|9] -> the reference's char cell arrays [n, rows*cols] (int8)"""
    from mbelib_neo_amd.layout import FRAME_CELLS, ROW_WIDTHS

    rows, cols = FRAME_CELLS[codec]
    bits = np.unpackbits(np.ascontiguousarray(frames, dtype=np.uint8), axis=1)
    cells = np.zeros((frames.shape[0], rows, cols), dtype=np.int8)
    off = 0
    for r, w in enumerate(ROW_WIDTHS[codec]):
        cells[:, r, :w] = bits[:, off:off + w][:, ::-1]   # the first wire bit of a row is its highest cell
        off += w
    return cells.reshape(frames.shape[0], rows * cols)


# ---- CPU baselines ------------------------------------------------------------------------------------------------
def _ref_lib(simd):
    import ctypes as C

    path = os.path.join(ROOT, "oracle", "_ref", "libref_bench_simd.so" if simd else "libref_bench.so")
    if not os.path.exists(path):
        return None
    try:
        lib = C.CDLL(path)
    except OSError:
        return None
    lib.ref_process_batch_mt.restype = C.c_int
    lib.ref_process_batch_mt.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    lib.ref_bench_recipe.restype = C.c_double
    lib.ref_bench_recipe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def reference_full_path(name, codec, T, simd, threads, budget_s):
    """The REAL reference (oracle/_ref: its own sources built -O2 IEEE by oracle/Makefile, scalar or with its SSE2 / PFFFT-SIMD
    paths) through oracle/tools/ref_bench.c on a bounded sample of the same workload.  None when the library is absent."""
    from mbelib_neo_amd.layout import FRAME_CELLS, init_state

    lib = _ref_lib(simd)
    if lib is None or "_soft" in name:
        return None
    per = (4096 if threads == 1 else 1024) if T == 1 else max(64, 4096 // T)
    S = per * threads
    cells = np.ascontiguousarray(unpack_cells(codec, make_frames(name, codec, S, T, rank=0)))
    ncell = FRAME_CELLS[codec][0] * FRAME_CELLS[codec][1]
    state = np.ascontiguousarray(init_state(S))
    pcm = np.zeros((S * T, 160), dtype=np.int16)

    def once():
        rc = lib.ref_process_batch_mt(codec, S, T, cells.ctypes.data, ncell, state.ctypes.data, 1234, pcm.ctypes.data, threads)
        assert rc == 0, rc

    once()   # warm-up pass (also warms the state, like the GPU run)
    done, t0 = 0, time.perf_counter()
    while True:
        once()
        done += S * T
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    build = "SSE2 + PFFFT SIMD (MBELIB_ENABLE_SIMD=1)" if simd else "scalar (SIMD off)"
    return {
        "value": done / dt,
        "unit": "frames/s",
        "cores": threads,
        "kind": "reference",
        "build": "simd" if simd else "scalar",
        "recipe": "full path (mbe_process*Frame, int16 out) on the bench workload",
        "sample": f"{done} frames ({S} streams x T={T}, same generator as the GPU workload) in {dt:.1f} s on {threads} thread(s), "
                  f"the reference's own sources built -O2 IEEE, {build}",
    }


def reference_recipe(recipe, simd):
    """bench/bench_synth.c:40-67 (recipe 0) / bench/bench_unvoiced.c:33-52,87-100 (recipe 1) / bench/bench_convert.c:33-50
    (recipe 2) of the reference, as functions of oracle/tools/ref_bench.c: 2,000 frames of mbe_synthesizeSpeechf per run,
    best of 5 (200,000 conversions, best of 3, for bench_convert), one core."""
    import ctypes as C

    lib = _ref_lib(simd)
    if lib is None:
        return None
    frames, runs = (2000, 5) if recipe < 2 else (200000, 3)
    sink = C.c_float()
    lib.ref_bench_recipe(recipe, 200, 1, C.byref(sink))
    best = lib.ref_bench_recipe(recipe, frames, runs, C.byref(sink))
    return {
        "value": frames / best,
        "unit": "frames/s",
        "cores": 1,
        "kind": "reference",
        "build": "simd" if simd else "scalar",
        "recipe": ("bench_synth (L=40, mixed voicing, w0 alternating)", "bench_unvoiced (L=36, all unvoiced)",
                   "bench_convert (mbe_floattoshort on one 160-sample ramp, bench/bench_convert.c:33-50)")[recipe],
        "us_per_frame": best / frames * 1e6,
        "sample": f"{frames} frames of {'mbe_synthesizeSpeechf' if recipe < 2 else 'mbe_floattoshort'}, best of {runs} runs ({best * 1e3:.1f} ms), single thread",
    }


def oracle_port_baseline(name, codec, T, budget_s):
    """the CPU oracle (oracle/mbx_oracle.c, a port of the reference path) when the reference build has not travelled"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    o = oracle_lib.load()
    soft = "_soft" in name
    S = (512 if soft else 4096) if T == 1 else max(64, 4096 // T)
    frames = make_frames(name, codec, S, T, rank=0)
    seeds = np.arange(S) + 1234
    state, rng = o.init_state(S), o.rng_seeded(seeds)
    out = o.process_batch(codec, S, T, frames, state, rng, soft=soft)  # warm-up pass (also warms the state)
    state, rng = out["state"], out["rng"]
    done, t0 = 0, time.perf_counter()
    while True:
        out = o.process_batch(codec, S, T, frames, state, rng, soft=soft)
        state, rng = out["state"], out["rng"]
        done += S * T
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {
        "value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port", "build": "scalar",
        "recipe": "full path on the bench workload",
        "sample": f"{done} frames ({S} streams x T={T}, same generator as the GPU workload) in {dt:.1f} s, single thread, "
                  "oracle/mbx_oracle.c -O2 IEEE",
    }


def host_cpu_share():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each
    GPU a share of the host's cores; threads beyond it only time-slice)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baselines(name, codec, T, full=True):
    """All CPU numbers of the line.  The first entry of the returned list is `cpu_baseline`: what north_star names -- the
    reference's SIMD build on the host cores of this box (all of them, count stated) -- or, without the reference
    build, the oracle port on one core.  Total budget about 25 s."""
    ncpu = host_cpu_share()
    out = []
    main = reference_full_path(name, codec, T, simd=True, threads=ncpu, budget_s=4.0)
    if main is None:
        main = reference_full_path(name, codec, T, simd=False, threads=ncpu, budget_s=4.0)
    if main is None:
        return [oracle_port_baseline(name, codec, T, 10.0)], ncpu
    out.append(main)
    if full:
        for simd, threads in ((True, 1), (False, 1), (False, ncpu)):
            r = reference_full_path(name, codec, T, simd=simd, threads=threads, budget_s=3.0)
            if r is not None:
                out.append(r)
        for recipe in (0, 1, 2):
            for simd in (True, False):
                r = reference_recipe(recipe, simd)
                if r is not None:
                    out.append(r)
    return out, ncpu


# ---- multi-GPU self-launch ------------------------------------------------------------------------------------------
def self_launch(argv, n, script=None):
    """Start the n ranks as child processes of this (GPU-free) process and relay rank 0's output.  Nothing here imports
    torch or touches HIP; a process that has initialised the GPU is never replaced."""
    script = script or os.path.abspath(__file__)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(rcs) if min(rcs) >= 0 else 1


# ---- one workload on this rank's GPU ----------------------------------------------------------------------------------
def parity_vs_oracle(name, codec, S, T, frames, seeds, launches, dec, out, step, streams=264):
    """CHECKER, after the timed region (never inside it, never the thing measured): a strided sample of the streams of the workload
    that was just timed, replayed through the CPU oracle (oracle/, the restatement pinned on the reference's golden vectors) over
    the SAME history -- every launch the decoder has seen since its construction, warm-up and timed steps alike -- and compared
    with what the device left: the mbe_process_result and int16 PCM of the LAST TIMED step, the float PCM of one more (untimed)
    step, and the final state.  This is the second half of BASELINE.json's metric ("+ PCM RMS error vs reference"); the tolerances
    are the ones tests/parity.py states (ref tests/test_golden_pcm.c:67-211 holds the reference's own golden-PCM check)."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import parity as par
    from mbelib_neo_amd import decoder
    from mbelib_neo_amd.layout import FRAME_BYTES, RESULT_DTYPE

    fb = FRAME_BYTES[codec]
    pick = np.arange(S // (2 * streams), S, max(1, S // streams))[:streams]
    d_pick = torch.from_numpy(pick).to(out["pcm16"].device)
    last16 = out["pcm16"].reshape(S, T, 160)[d_pick].cpu().numpy()
    last_res = out["results"].reshape(S, T, 5)[d_pick].cpu().numpy()
    extra = dec.make_outputs(T, want_pcm16=True, want_float=True, want_results=True)
    t0 = time.perf_counter()
    dec.decode(frames, T, out=extra)        # one more launch of the same batch, float PCM on
    torch.cuda.synchronize()
    pf = extra["pcmf"].reshape(S, T, 160)[d_pick].cpu().numpy()
    p16 = extra["pcm16"].reshape(S, T, 160)[d_pick].cpu().numpy()
    state = dec.state_numpy()[pick]
    o = oracle_lib.load()
    n = launches + 1
    fr = np.asarray(frames).reshape(S, T, fb)[pick]
    hist = np.ascontiguousarray(np.tile(fr, (1, n, 1))).reshape(-1, fb)    # the same T frames per launch, n launches
    ref = o.process_batch(codec, len(pick), n * T, hist, o.init_state(len(pick)), o.rng_seeded(seeds[pick]))
    r16 = ref["pcm16"].reshape(len(pick), n, T, 160)
    rf = ref["pcmf"].reshape(len(pick), n, T, 160)
    rres = ref["results"].reshape(len(pick), n, T)
    violations = []
    results_exact = True
    try:
        par.check_results(rres[:, n - 2].reshape(-1), np.ascontiguousarray(last_res).view(RESULT_DTYPE).reshape(-1), what="last timed step")
    except AssertionError as e:
        results_exact = False
        violations.append("results: " + str(e)[:200])
    # PCM error of the untimed extra step (float) and of the last timed step + the extra step (int16).  Every bound is tests/parity.py's: the
    # float criteria, and per frame int16_bound(the frame's pre-clip peak, which the oracle reports) -- the headline workload repeats one clean
    # all-voiced frame per stream tick after tick and sits in the soft clip in about half of its frames by construction, so the bound that is
    # relative to the amplitude a frame was computed at is the one that applies; nothing here is looser than what the test suite holds.
    rel, worst, _ = par.pcm_float_stats(rf[:, n - 1], pf)
    peaks = ref["peak"].reshape(len(pick), n, T)
    st_last, bad_last = par.int16_stats(rf[:, n - 2], r16[:, n - 2], last16, peak=peaks[:, n - 2])
    st_extra, bad_extra = par.int16_stats(rf[:, n - 1], r16[:, n - 1], p16, peak=peaks[:, n - 1])
    if rel > par.PCM_REL_RMS:
        violations.append(f"PCM relative RMS {rel:.3e} > {par.PCM_REL_RMS:.1e}")
    if worst > par.PCM_WORST_FRAME:
        violations.append(f"worst frame {worst:.3e} > {par.PCM_WORST_FRAME:.1e}")
    violations += ["last timed step: " + b for b in bad_last] + ["extra step: " + b for b in bad_extra]
    state_ok = True
    try:
        par.check_state(ref["state"], state)
    except AssertionError as e:
        state_ok = False
        violations.append("state: " + str(e)[:200])
    return {   # a bound that does not hold is REPORTED in the object ("FAILED"), next to the numbers, not instead of them
        **({"FAILED": "bench parity: " + "; ".join(violations)} if violations else {}),
        "rel_rms": rel, "worst_frame": worst,
        "int16_within_1": st_last["int16_within_1"], "int16_exact": st_last["int16_exact"],
        "int16_max": max(st_last["int16_max"], st_extra["int16_max"]),
        "int16_max_below_clip": max(st_last["int16_max_below_clip"], st_extra["int16_max_below_clip"]),
        "int16_max_inside_clip": max(st_last["int16_max_inside_clip"], st_extra["int16_max_inside_clip"]),
        "int16_margin": min(st_last["int16_margin"], st_extra["int16_margin"]),   # smallest (bound - difference) over the checked frames
        "int16_bound": st_last["int16_bound"], "clipped_frames": st_last["clipped_frames"],
        "preclip_peak_max": float(max(peaks[:, n - 2].max(), peaks[:, n - 1].max())),
        "streams_checked": int(len(pick)), "launches_replayed": int(n), "frames_checked": int(len(pick) * T),
        "results_exact": results_exact, "state_in_tolerance": state_ok,
        "tolerance": {"rel_rms": par.PCM_REL_RMS, "worst_frame": par.PCM_WORST_FRAME, "int16": "tests/parity.py int16_bound(pre-clip peak)",
                      "int16_within_1_share": 0.999},
        "oracle": "oracle/mbx_oracle.c (CPU restatement pinned on the reference's golden vectors; double-precision FFT form)",
        "what": "strided sample of the timed workload's streams replayed through the oracle over every launch since the decoder's "
                "construction: results + int16 PCM of the last timed step, float PCM of one more untimed step, final state",
        "seconds": time.perf_counter() - t0,
    }


def run_workload(name, S, T, steps, warmup, rank, first_stream, local_rank, blob, world, dist, args, use_dist=False, coll_device=None,
                 alternate=False, min_time_s=None, serial=None, parity=False):
    """Returns the measurements of one workload: wall time of the timed steps (max over ranks), HIP-event time of the
    dominant kernel per step (mean, median, spread), frame mix.  The launches of a step are what mbx_process_batch issues;
    they are issued one by one here only so that the dominant kernel can be bracketed by events on the launch stream.

    alternate = False (every reported headline): every launch walks the streams in the same direction, so a launch finds
    nothing of the previous launch's state in the Infinity Cache -- what a decoder that ticks every 20 ms with other work
    in between sees, and the timing the HBM roofline fraction is quoted on.  alternate = True is the library's default
    order for back-to-back launches (reported separately as `infinity_cache_assisted`).

    The timed region is at least `steps` steps and at least `min_time_s` of wall time (steps_effective, the same on
    every rank): a 20-step region of a 0.28 ms step is 5.6 ms, too short to resolve effects of a few per cent."""
    import torch

    from mbelib_neo_amd import _native, decoder

    codec = WORKLOADS[name][0]
    resident = name.endswith("_resident")
    frames = make_frames(name, codec, S, T, rank)
    dec = decoder.BatchDecoder(codec, S, device=local_rank, seeds=np.arange(first_stream, first_stream + S) + 1234, tables_blob=blob,
                               resident=resident)
    d_frames = dec.to_device(frames)
    out = dec.make_outputs(T, want_pcm16=True, want_float=False, want_results=True)
    L = _native.lib()
    if args.ablate and not hasattr(L, "mbx_debug_set_ablation"):
        raise SystemExit("--ablate needs the development build: make -C mbelib-neo_amd/csrc ablate && "
                         "MBX_HIP_LIBRARY=$PWD/mbelib-neo_amd/libmbx_hip_ablate.so python bench.py --ablate MASK")
    stream = torch.cuda.current_stream().cuda_stream
    _native.check(L.mbx_reserve_stream(stream, S * T), "mbx_reserve_stream")  # launches below never allocate
    soft = "_soft" in name
    if soft:
        def fec(frames_ptr, count, records_ptr, strm):
            return L.mbx_fec_soft(codec, frames_ptr, count, records_ptr, strm)
    else:
        fec = {0: L.mbx_fec_imbe7200x4400, 1: L.mbx_fec_ambe3600x2450, 2: L.mbx_fec_imbe7100x4400, 3: L.mbx_fec_ambe3600x2450}[codec]
    n = S * T
    stream_codec = 0 if codec == 2 else codec   # 7100x4400 records are in 7200x4400 order after its FEC stage
    # mbx_process_records is ONE launch where the stream kernel expands the records itself (IMBE at T > 1, the AMBE codecs
    # at T >= 4); otherwise it is the expand launch + the stream launch, issued separately here so that the events bracket
    # the stream kernel alone.  --split-expand forces the separate launch everywhere (development aid).
    split = bool(L.mbx_uses_expand_launch(stream_codec, S, T)) or args.split_expand
    # IMBE at T = 1: mbx_process_batch is ONE launch (imbe_stream_kernel_one_fused: FEC + expansion + stream stage in the stream's
    # own wave) unless MBX_FUSE_ONE=0; the step is then that call, and the events bracket it.
    batch_kernel = L.mbx_batch_kernel_name(codec, S, T, 1 if resident else 0).decode()
    fused = (not soft) and (batch_kernel.endswith("_fused") or "one_launch" in batch_kernel) and not args.split_expand
    # Front-end overlap (round 4).  FEC + parameter expansion of a batch depend only on its frames; the stream stage of the batch
    # before it depends only on ITS records / rows and on the state.  So the front end of step k + 1 is issued on a second HIP
    # stream into alternating record / workspace buffers (the *_ws entry points) and runs while the stream kernel of step k does;
    # events order the two: the stream stage waits for its front end, a front end waits until the stream stage that last read its
    # buffers is done.  Every step still does all of its work inside the timed region.  Opt-in (--overlap-front-end): it measured
    # slower than one stream -- the HBM-bound stream kernel does not like company, and two event hand-overs per step cost ~20 us.
    serial = (not args.overlap_front_end) if serial is None else serial
    overlap = split and not soft and not serial
    if overlap:
        front = torch.cuda.Stream(device=local_rank)
        main_stream = torch.cuda.current_stream()
        ws_bytes = int(L.mbx_workspace_bytes(n))
        bufs = [{"records": torch.empty_like(out["records"]), "ws": torch.empty(ws_bytes, dtype=torch.uint8, device=out["records"].device),
                 "ready": torch.cuda.Event(), "free": torch.cuda.Event()} for _ in range(2)]
        counter = [0]

        def step(ev=None):
            k = counter[0]
            counter[0] += 1
            b = bufs[k & 1]
            if k >= 2:
                front.wait_event(b["free"])          # the stream stage of step k - 2 has finished with these buffers
            _native.check(fec(d_frames.data_ptr(), n, b["records"].data_ptr(), front.cuda_stream), "fec")
            _native.check(L.mbx_expand_records_ws(stream_codec, b["records"].data_ptr(), n, b["ws"].data_ptr(), ws_bytes, front.cuda_stream),
                          "expand")
            b["ready"].record(front)
            main_stream.wait_event(b["ready"])
            if ev is not None:
                ev[0].record()
            _native.check(
                L.mbx_stream_expanded_ws(stream_codec, S, T, b["records"].data_ptr(), dec.state.data_ptr(),
                                         dec.resident.data_ptr() if resident else None, dec.rng.data_ptr(), out["pcm16"].data_ptr(), None,
                                         out["results"].data_ptr(), b["ws"].data_ptr(), ws_bytes, stream),
                "stream",
            )
            if ev is not None:
                ev[1].record()
            b["free"].record(main_stream)
    def step_fused(ev=None):
        if ev is not None:
            ev[0].record()
        if resident:
            rc = L.mbx_process_batch_resident(codec, S, T, None, d_frames.data_ptr(), dec.state.data_ptr(), dec.resident.data_ptr(),
                                              dec.rng.data_ptr(), out["pcm16"].data_ptr(), None, out["results"].data_ptr(),
                                              out["records"].data_ptr(), stream)
        else:
            rc = L.mbx_process_batch(codec, S, T, d_frames.data_ptr(), dec.state.data_ptr(), dec.rng.data_ptr(), out["pcm16"].data_ptr(),
                                     None, out["results"].data_ptr(), out["records"].data_ptr(), stream)
        _native.check(rc, "mbx_process_batch")
        if ev is not None:
            ev[1].record()

    def step_serial(ev=None):
        if ev is not None and soft:
            ev[0].record()
        _native.check(fec(d_frames.data_ptr(), n, out["records"].data_ptr(), stream), "fec")
        if ev is not None and soft:
            ev[1].record()
        run = L.mbx_process_records
        if split:
            _native.check(L.mbx_expand_records(stream_codec, out["records"].data_ptr(), n, stream), "expand")
            run = L.mbx_stream_expanded
        if ev is not None and not soft:
            ev[0].record()
        if resident:
            assert split, "the resident bench workloads are the T = 1 ones (expand launch + stream launch)"
            _native.check(
                L.mbx_stream_expanded_resident(stream_codec, S, T, out["records"].data_ptr(), dec.state.data_ptr(), dec.resident.data_ptr(),
                                               dec.rng.data_ptr(), out["pcm16"].data_ptr(), None, out["results"].data_ptr(), stream),
                "stream (resident)",
            )
        else:
            _native.check(
                run(stream_codec, S, T, out["records"].data_ptr(), dec.state.data_ptr(), dec.rng.data_ptr(),
                    out["pcm16"].data_ptr(), None, out["results"].data_ptr(), stream),
                "stream",
            )
        if ev is not None and not soft:
            ev[1].record()

    if fused:
        step = step_fused
        overlap = False
    elif not overlap:
        step = step_serial
    launches = [0]
    inner_step = step

    def step(ev=None):   # noqa: F811 -- counts the launches the state has seen (the parity replay needs the whole history)
        launches[0] += 1
        inner_step(ev)
    previous_order = L.mbx_set_stream_order(1 if alternate else 0)
    for _ in range(max(1, warmup)):  # the first pass also warms the model state
        step()
    if args.ablate:
        L.mbx_debug_set_ablation(args.ablate)  # development build only (tools/); never in a reported run
    min_time_s = args.min_time_ms * 1e-3 if min_time_s is None else min_time_s
    steps_eff = steps
    if min_time_s > 0:   # size the timed region from three untimed steps (part of the warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        est = (time.perf_counter() - t0) / 3
        steps_eff = max(steps, int(np.ceil(min_time_s / max(est, 1e-6))))
        if use_dist:
            t = torch.tensor([steps_eff], dtype=torch.int64, device=coll_device if coll_device is not None else torch.device("cuda", local_rank))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steps_eff = int(t[0])
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps_eff)]
    # Load-onset transient (profiles/r05/launch_series.txt): after an idle gap the first ~6 launches run at the steady rate, the next
    # ~100 up to 45 % slower (the power controller clamps, then relaxes), and only then the kernel time is what a decoder that runs tick
    # after tick sees.  Building the events above IS such a gap, so the warm-up ends with `settle` more untimed steps (~30 ms of them)
    # right in front of the timed region, and the timed region is long (--min-time-ms) against what the barrier's own gap re-triggers.
    settle = 0
    if min_time_s > 0:
        settle = int(np.ceil(min(0.030, min_time_s) / max(est, 1e-6)))
        for _ in range(settle):
            step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps_eff):
        step(events[k])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt_local = dt = time.perf_counter() - t0
    per_step = np.array([a.elapsed_time(b) for a, b in events])
    if getattr(args, "dump_steps", None):   # diagnosis: the event-bracketed time of every timed launch, in order
        np.savetxt(args.dump_steps, per_step, fmt="%.5f")
    kernel_ms = float(per_step.mean())
    if use_dist:
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device=coll_device if coll_device is not None else torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms = float(t[0]), float(t[1])
    L.mbx_set_stream_order(previous_order)
    hist = decoder.result_histogram(out["results"], stream)   # the last launch's tally of flags, formed on the device (mbx_result_histogram)
    pcm_digest = int(out["pcm16"].to(torch.int64).sum().item())
    parity_obj = None
    if parity and not soft and not overlap and not args.ablate:
        try:
            parity_obj = parity_vs_oracle(name, codec, S, T, frames, np.arange(first_stream, first_stream + S) + 1234, launches[0], dec, out, step,
                                          streams=264 if T <= 4 else 64)
        except AssertionError as e:   # a parity failure is reported in the line, loudly, not swallowed
            parity_obj = {"FAILED": str(e)[:400]}
    if soft:
        kernel = {0: "fec_imbe7200x4400_soft_kernel", 1: "fec_ambe3600x2450_soft_kernel", 2: "fec_imbe7100x4400_soft_kernel",
                  3: "fec_ambe3600x2450_soft_kernel"}[codec]
        alg_bytes = soft_fec_bytes_per_launch(codec, n)
    else:
        kernel = batch_kernel   # the dominant kernel of the step for this shape (one-launch / time-sliced / plain instance)
        alg_bytes = algorithmic_bytes_per_launch(codec, S, T, resident)
    del dec, d_frames, out
    torch.cuda.empty_cache()
    return {
        "dt": dt, "kernel_ms": kernel_ms, "kernel": kernel, "alg_bytes": alg_bytes, "frames_per_step": world * n,
        "value": world * n * steps_eff / dt, "ms_per_step": dt / steps_eff * 1e3, "pcm_digest": pcm_digest, "front_end_overlap": bool(overlap),
        "steps_effective": steps_eff, "settle_steps": settle, "rank_value": n * steps_eff / dt_local, "parity": parity_obj,
        "kernel_ms_stats": {"mean": float(per_step.mean()), "median": float(np.median(per_step)), "p10": float(np.percentile(per_step, 10)),
                            "p90": float(np.percentile(per_step, 90)), "min": float(per_step.min()), "max": float(per_step.max()),
                            "launches": int(per_step.size)},
        "frame_mix": {k: hist[k] / max(hist["frames"], 1) for k in ("repeat", "mute", "erasure", "tone")},
    }


def convert_rate(local_rank, nframes=1 << 20, steps=20):
    """floattoshort_kernel (mbe_floattoshort for a batch, ref src/core/mbelib.c:1148-1321; the reference's bench_convert
    recipe is its CPU counterpart in cpu_baselines): frames/s and the HBM fraction of 960 B per frame."""
    import torch

    from mbelib_neo_amd import _native

    L = _native.lib()
    dev = torch.device("cuda", local_rank)
    x = (torch.rand((nframes, 160), device=dev) - 0.5) * 12000.0
    y = torch.empty((nframes, 160), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _native.check(L.mbx_floattoshort(x.data_ptr(), y.data_ptr(), nframes, stream), "mbx_floattoshort")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        _native.check(L.mbx_floattoshort(x.data_ptr(), y.data_ptr(), nframes, stream), "mbx_floattoshort")
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    return {"kernel": "floattoshort_kernel", "frames_per_launch": nframes, "kernel_ms": ms, "frames_per_s": nframes / (ms * 1e-3),
            "bytes_per_frame": 960, "frac": nframes * 960 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "what": "float PCM resident in HBM -> int16 PCM, 1,048,576 frames per launch (the device counterpart of the reference's bench_convert)"}


def copy_floor(local_rank, S=65536, min_time_s=0.25):
    """state_copy_kernel: every stream's three structs loaded and stored with exactly the stream kernels' accesses (one wave per
    stream, a dword per lane) and nothing else -- what the chip gives THIS access pattern, timed in this very run.  The T = 1
    stream kernels are priced against it beside the 8 TB/s datasheet roofline (roofline.copy_floor).
    Timed like the headline: ~30 ms of untimed launches first, then at least `min_time_s` of launches between two events -- twenty
    launches behind an idle gap (how rounds 1-5 timed it) sit in the load-onset transient of the power controller and read 8 % low
    (5.4 instead of 5.9 TB/s; tools/copy_patterns2.hip, profiles/r06/copy_patterns2.json)."""
    import torch

    from mbelib_neo_amd import _native, decoder

    L = _native.lib()
    dec = decoder.BatchDecoder(0, S, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def run(n):
        for _ in range(n):
            _native.check(L.mbx_state_copy(S, dec.state.data_ptr(), stream), "mbx_state_copy")

    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(20)
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 20, 1e-6)
    n = max(20, int(min_time_s / est))
    run(max(1, int(0.030 / est)))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run(n)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    nbytes = 2 * S * 3 * 2604
    return {"kernel": "state_copy_kernel", "bytes_per_launch": nbytes, "kernel_ms": ms, "rate_GBps": nbytes / (ms * 1e-3) / 1e9, "launches_timed": n,
            "what": "load + store of the three structs of 65,536 streams with the stream kernels' access pattern, no arithmetic; steady state"}


def roofline_of(name, S, T, m):
    traffic, source = measured_traffic(name, S, T)
    achieved = m["alg_bytes"] / (m["kernel_ms"] * 1e-3) / 1e9
    r = {
        "bound": "hbm",
        "kernel": m["kernel"],
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": source,
        "algorithmic_bytes_per_launch": m["alg_bytes"],
        "kernel_ms": m["kernel_ms"],
        "kernel_ms_stats": m["kernel_ms_stats"],
        "stream_order": "fixed: every launch walks the streams forward, nothing of the previous launch's state is found in the Infinity Cache",
        "frac_on_counter_bytes": (traffic / (m["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "issue": measured_issue(name, S, T),
    }
    if "_soft" in name:
        r["note"] = "dominant kernel of this workload is the soft-decision FEC kernel: algorithmic bytes = n * (2 B per soft cell + 16 B record)"
    else:
        r["note"] = "algorithmic bytes = S*T*(wire frame + int16 PCM) + S*2*3*2604 state (SURVEY.md §8(d))"
    r["dispatches_per_step"] = 1
    if m["kernel"].endswith("_slice"):
        # a sliced launch (include/mbx.h): the step is 3 groups of streams x ceil(T / slice) slices, issued on three HIP queues
        # and overlapping on the device; kernel_ms brackets the whole step on the caller's stream (fork event to join events), a
        # rocprofv3 AverageNs of `kernel` is ONE slice of one group.  `traffic` and the issue model are per STEP: sums over the
        # step's dispatches (tools/profile_round.sh, tools/sq_profile.py)
        from mbelib_neo_amd import _native
        tc = int(_native.lib().mbx_launch_slices(WORKLOADS[name][0], S, T))
        r["dispatches_per_step"] = 3 * ((T + tc - 1) // tc) if tc > 0 else 1
        r["slice_frames"] = tc
        r["note"] += ("; SLICED launch: kernel_ms is the whole step (concurrent slices on three queues), one rocprofv3 dispatch of "
                      "this kernel is one slice of one group of streams; traffic / issue = sums over the step's dispatches")
    r["traffic_over_algorithmic"] = (traffic / m["alg_bytes"]) if traffic else None
    return r


LINE_LIMIT = 4096   # bytes: the driver keeps only the tail of stdout, and a 20 KB line (round 5) was not parsed


def _sig(x, digits=6):
    """floats of the compact line to `digits` significant digits (ints, strings, None untouched)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{digits}g}")


def _pick(d, keys, digits=6):
    return {k: _sig(d[k], digits) for k in keys if d is not None and k in d}


def contract_line(detail):
    """The ONE stdout line: the bench contract's keys plus `roofline`, `parity`, `cpu_baseline` and a five-number summary per
    other config.  `detail` is the full measurement dict (what bench_detail.json holds).  Pure (no torch, no GPU): covered by a CPU
    test on a canned dict.  Never longer than LINE_LIMIT bytes: free-text fields are clipped first, and should a future key push it
    over, `other_configs` and then the free text are dropped before anything the contract names."""
    line = _pick(detail, ("metric", "value", "unit", "n_gpus", "steps", "steps_effective", "warmup", "ms_per_step", "higher_is_better",
                          "scaling", "vs_baseline", "dtype", "data"), digits=7)
    cfg = detail.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "streams_per_gpu", "frames_per_stream_per_step", "frames_per_step", "output", "parallelism"))
    rf = detail.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                  "traffic_over_algorithmic", "algorithmic_bytes_per_launch", "kernel_ms", "dispatches_per_step"))
    if line["roofline"].get("achieved") and line["roofline"].get("peak"):   # self-consistent after the rounding: frac IS achieved / peak
        line["roofline"]["frac"] = line["roofline"]["achieved"] / line["roofline"]["peak"]
    par = detail.get("parity")
    if par is not None:
        line["parity"] = _pick(par, ("FAILED", "rel_rms", "worst_frame", "int16_within_1", "int16_max", "int16_bound", "streams_checked",
                                     "launches_replayed", "results_exact", "state_in_tolerance"), digits=4)
    cb = detail.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "build", "sample"))
    oc = detail.get("other_configs")
    if oc:
        line["other_configs"] = {k: _pick(v, ("value", "ms_per_step", "kernel", "kernel_ms", "frac"), digits=5) for k, v in oc.items()}
    if detail.get("detail_file"):
        line["detail"] = detail["detail_file"]

    def size():
        return len(json.dumps(line))

    for obj, key, keep in ((line["config"], "workload", 160), (line.get("cpu_baseline") or {}, "sample", 200), (line["roofline"], "traffic_source", 80),
                           (line.get("parity") or {}, "FAILED", 300), (line["config"], "parallelism", 80)):
        if size() > LINE_LIMIT and isinstance(obj.get(key), str) and len(obj[key]) > keep:
            obj[key] = obj[key][:keep - 3] + "..."
    for victim in ("other_configs", "detail"):
        if size() > LINE_LIMIT:
            line.pop(victim, None)
    assert size() <= LINE_LIMIT, size()
    return line


def write_detail(detail):
    """the sidecar: next to this script, and under gpurun_out/ (merged back from the GPU box) when that directory exists"""
    written = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(detail, f, indent=1)
                written.append(os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT))
            except OSError:
                pass
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="imbe_voiced", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle replay of a sample of the timed workload (the line's `parity` object)")
    ap.add_argument("--no-extras", action="store_true", help="headline (and cpu_baseline) only: no other_configs, host_path, infinity_cache_assisted")
    ap.add_argument("--split-expand", action="store_true", help="run the parameter expansion as a separate launch")
    ap.add_argument("--overlap-front-end", action="store_true",
                    help="development aid: FEC + expansion of step k + 1 on a second stream while the stream stage of step k runs "
                         "(mbx_*_ws entry points).  Measured SLOWER than one stream (imbe_voiced 255 vs 265 M frames/s on one box: the "
                         "stream kernel shares the chip, and the cross-stream events cost what the overlap saves) -- not the default")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the collectives even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collectives backend; gloo (CPU tensors) only to rehearse N > 1 on a box with fewer GPUs than ranks, "
                         "together with MBX_BENCH_SHARE_GPU=1 (every rank on device 0)")
    ap.add_argument("--ablate", type=int, default=0, help="timing-only stage mask (development build of the library only); results invalid")
    ap.add_argument("--dump-steps", default=None, help="diagnosis: write the event-bracketed time (ms) of every timed step, in order, to this file")
    ap.add_argument("--min-time-ms", type=float, default=500.0,
                    help="the timed region is at least --steps steps AND at least this much wall time (steps_effective in the line)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one, before anything touches a GPU
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MBX_BENCH_SHARE_GPU"):   # rehearsal only: all ranks on one card (RCCL refuses that: use --dist-backend gloo)
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_device = device if args.dist_backend == "nccl" else torch.device("cpu")   # where the tensors of the collectives live

    import mbelib_neo_amd as mbx
    from mbelib_neo_amd import decoder
    from mbelib_neo_amd.parallel import broadcast_tables, shard_range

    # rank 0 reads the blob, RCCL broadcasts it, every rank uploads it and the checksums are compared
    blob = broadcast_tables(mbx.load_tables_blob() if rank == 0 else None, coll_device, force=args.force_dist)
    checksum = decoder.ensure_init(local_rank, blob)
    checksums = [checksum]
    if use_dist:
        cs = torch.tensor([checksum], dtype=torch.int64, device=coll_device)
        gathered = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        checksums = [int(g) for g in gathered]
        assert all(g == checksum for g in checksums), "table checksum differs between ranks"

    codec, S, T, desc = WORKLOADS[args.workload]
    if args.streams:
        S = args.streams
    first, count = shard_range(S * world, world, rank)  # weak scaling: S streams on every rank
    assert count == S
    m = run_workload(args.workload, S, T, args.steps, args.warmup, rank, first, local_rank, blob, world, dist, args, use_dist, coll_device,
                     parity=(rank == 0 and not args.no_parity and not args.no_extras))
    rank_values = [m["rank_value"]]
    if use_dist:   # what every rank measured on its own clock (the line's value is the max-over-ranks time)
        rv = torch.tensor([m["rank_value"]], dtype=torch.float64, device=coll_device)
        gathered = [torch.zeros_like(rv) for _ in range(world)]
        dist.all_gather(gathered, rv)
        rank_values = [float(g) for g in gathered]

    line = {
        "metric": "20ms frames/sec (whole node) + PCM RMS error vs reference, " + CODEC_NAME[codec],
        "value": m["value"],
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "steps_effective": m["steps_effective"],
        "settle_steps": m["settle_steps"],   # untimed steps directly in front of the timed region, beyond --warmup
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.ablate else f"synthetic, ABLATED stages mask={args.ablate} (timing only, INVALID as a result)",
        "config": {
            "workload": desc,
            "streams_per_gpu": S,
            "frames_per_stream_per_step": T,
            "frames_per_step": m["frames_per_step"],
            "output": "int16 PCM + mbe_process_result per frame",
            "parallelism": f"{world} independent stream shard(s), table blob broadcast over RCCL, per-rank checksums equal"
                           + (" (process group initialised)" if use_dist else ""),
            "frame_mix": m["frame_mix"],
        },
        "roofline": roofline_of(args.workload, S, T, m),
        # the metric's second half: PCM error of the timed workload against the CPU oracle (a strided sample of its streams; checker only)
        "parity": m["parity"],
        "distributed": {
            "process_group": (dist.get_backend() if use_dist else None),
            "world_size": (dist.get_world_size() if use_dist else 1),
            "devices_visible": torch.cuda.device_count(),
            "per_rank_frames_per_s": rank_values,
            "table_checksums": checksums,
            "tables": "rank 0 reads the blob, one broadcast over the process group, every rank uploads its copy" if use_dist else "read and uploaded by the only rank",
        },
    }
    fpf, fpf_basis = nominal_flops_per_frame(args.workload)
    line["valu"] = {   # "also report valu.achieved"; the resource that binds this path
        "achieved": m["value"] / world * fpf / 1e12,
        "peak": VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s per GPU",
        "frac": m["value"] / world * fpf / 1e12 / VALU_PEAK_TFLOPS,
        "flops_per_frame": fpf,
        "basis": fpf_basis,
    }
    issue = line["roofline"].get("issue") or {}
    # MODELLED from measured counts: VALU issue utilisation of the dominant kernel from the SQ counters of this build (instruction counts
    # priced with the per-instruction costs tools/valu_issue.hip measured; EXPERIMENTS.md, rounds 3 / 4), None without a matching summary
    line["valu"]["modelled_valu_issue_utilisation"] = issue.get("valu_issue_utilisation")               # every instruction at the cost of the expensive class
    line["valu"]["modelled_valu_issue_utilisation_lower_bound"] = issue.get("valu_issue_utilisation_lower_bound")
    line["valu"]["cost_model"] = issue.get("valu_cost_model")   # per-instruction costs from tools/valu_issue.hip at FOUR waves per SIMD; the kernels run at 3.8-7
    line["valu"]["modelled_source"] = issue.get("source") or issue.get("stale")
    line["front_end_overlap"] = m["front_end_overlap"]   # FEC + expansion of step k + 1 on a second stream while the stream stage of step k runs
    extras = world == 1 and not args.no_extras and not args.ablate and not args.streams
    if extras and m["front_end_overlap"]:
        # the same workload with the three launches of a step on ONE stream, one after the other (how rounds 1-3 timed it)
        sm = run_workload(args.workload, S, T, args.steps, 2, rank, first, local_rank, blob, world, dist, args, serial=True)
        line["serial_front_end"] = {
            "value": sm["value"], "ms_per_step": sm["ms_per_step"], "kernel_ms": sm["kernel_ms"], "steps_effective": sm["steps_effective"],
            "what": "FEC, expansion and stream stage of a step issued on one stream: the step is the sum of its three launches",
        }
    if extras:
        # the library's default order for back-to-back launches over the same state: a launch walks the streams in the
        # direction opposite to the previous one and finds the tail of its state in the 256 MiB Infinity Cache
        am = run_workload(args.workload, S, T, args.steps, 2, rank, first, local_rank, blob, world, dist, args, alternate=True)
        line["infinity_cache_assisted"] = {
            "value": am["value"], "ms_per_step": am["ms_per_step"], "kernel_ms": am["kernel_ms"], "kernel_ms_stats": am["kernel_ms_stats"],
            "frac": am["alg_bytes"] / (am["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "what": "mbx_set_stream_order(1), the library default: successive launches walk the streams in opposite directions; "
                    "NOT an HBM figure (part of the state comes out of the Infinity Cache), reported beside the headline only",
        }
    if extras and args.workload == "imbe_voiced":
        # the other three GPU configs of BASELINE.json, driver-timed in the same line (10 steps each)
        line["other_configs"] = {}
        for other in ("imbe_voiced_resident", "ambe_fec", "ambe_fec_resident", "imbe_mixed", "ambe_stream"):
            oc, oS, oT, odesc = WORKLOADS[other]
            om = run_workload(other, oS, oT, 10, 2, rank, 0, local_rank, blob, 1, dist, args)
            orf = roofline_of(other, oS, oT, om)
            line["other_configs"][other] = {
                "workload": odesc, "value": om["value"], "unit": "frames/s", "steps": 10, "steps_effective": om["steps_effective"],
                "ms_per_step": om["ms_per_step"],
                "kernel": om["kernel"], "kernel_ms": om["kernel_ms"], "kernel_ms_stats": om["kernel_ms_stats"],
                "algorithmic_bytes_per_launch": om["alg_bytes"],
                "frac": orf["frac"], "traffic": orf["traffic"], "frac_on_counter_bytes": orf["frac_on_counter_bytes"], "issue": orf["issue"],
                "frame_mix": om["frame_mix"],
            }
        try:
            line["convert"] = convert_rate(local_rank)
        except Exception as e:   # noqa: BLE001 -- the headline must not depend on the extras
            line["convert"] = {"error": str(e)[:300]}
        if T == 1:
            try:   # the same bytes through a kernel that only copies them: how far the stream kernel is from what this access pattern can reach
                cf = copy_floor(local_rank)
                rf = line["roofline"]
                moved = rf.get("traffic") or rf["algorithmic_bytes_per_launch"]
                cf["stream_kernel_bytes"] = moved
                cf["stream_kernel_bytes_source"] = "PMC counters of this build" if rf.get("traffic") else "algorithmic bytes"
                cf["stream_kernel_rate_GBps"] = moved / (rf["kernel_ms"] * 1e-3) / 1e9
                cf["frac_of_copy_rate"] = cf["stream_kernel_rate_GBps"] / cf["rate_GBps"]
                rf["copy_floor"] = cf
            except Exception as e:   # noqa: BLE001
                line["roofline"]["copy_floor"] = {"error": str(e)[:300]}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # timed at N = 1 only, on rank 0
            lst, ncpu = cpu_baselines(args.workload, codec, T, full=extras)
            line["cpu_baseline"] = lst[0]
            line["cpu_baselines"] = lst
            line["host_cores"] = ncpu
        if extras and args.workload == "imbe_voiced":
            # what a C host sees from host memory to host memory (a child process: plain C against the two libraries)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import host_path_rate

                hp = json.loads(host_path_rate.run(65536, local_rank))
                sync_variants = {}
                for key, env in (("frame_server", {"MBE_NEO_FRAME_SERVER": "1"}), ("no_device_copy", {"MBE_NEO_FRAME_SHADOW": "0"})):
                    try:
                        sync_variants[key] = json.loads(host_path_rate.run(4096, local_rank, dict(env, MBX_HOST_BENCH_SYNC_ONLY="1")))["sync_call_us"]
                    except Exception as e:   # noqa: BLE001
                        sync_variants[key] = str(e)[:200]
                one_core = [b for b in line.get("cpu_baselines", []) if b["cores"] == 1 and b["recipe"].startswith("full path")]
                line["host_path"] = {
                    "frames_per_s": hp["session_pinned_frames_per_s"],
                    "bytes_over_pcie_per_frame": 18 + 320,
                    "what": "mbx_session_submit, 65,536 IMBE streams x T=1 per submit, 40 submits, pinned host buffers in and out "
                            "(int16 PCM), state resident on the device; end to end from host memory to host memory",
                    "with_results_frames_per_s": hp["session_pinned_with_results_frames_per_s"],
                    "pageable_buffers_frames_per_s": hp["session_pageable_frames_per_s"],
                    "frames_per_s_by_host_threads": dict(zip(map(str, hp.get("threads", [])), hp.get("session_pinned_frames_per_s_by_threads", []))),
                    "devices": hp.get("devices"), "all_devices_frames_per_s": hp.get("session_all_devices_frames_per_s"),
                    "per_frame_api": {
                        "sync_call_us": hp["sync_call_us"],
                        "sync_call_us_frame_server": sync_variants.get("frame_server"),       # MBE_NEO_FRAME_SERVER=1 (opt-in)
                        "sync_call_us_no_device_copy": sync_variants.get("no_device_copy"),   # MBE_NEO_FRAME_SHADOW=0
                        "reference_call_us": (1e6 / one_core[0]["value"]) if one_core else None,
                        "queue_mode_resident_frames_per_s": hp["queue_resident_frames_per_s"],
                        "queue_mode_writeback_frames_per_s": hp["queue_writeback_frames_per_s"],
                        "queue_mode_host_ns_per_call": hp["queue_resident_call_ns"],
                        "queue_channels": hp["queue_channels"],
                        "queue_mode_resident_frames_per_s_by_host_threads": dict(zip(map(str, hp.get("threads", [])), hp.get("queue_resident_frames_per_s_by_threads", []))),
                        "what": "mbe_processImbe7200x4400Frame through libmbe_neo_amd.so from one host thread: synchronous "
                                "(S=T=1 round trip per call) and in queue mode (mbe_batchBegin / mbe_flush, one frame per channel and flush)",
                    },
                }
            except Exception as e:   # noqa: BLE001 -- the headline must not depend on the extras
                line["host_path"] = {"error": str(e)[:300]}
        line["detail_file"] = "bench_detail.json"
        write_detail(line)
        print(json.dumps(contract_line(line)), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
