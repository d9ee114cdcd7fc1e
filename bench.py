#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched MBE decode path on MI355X.

Metric (BASELINE.json): 20 ms frames/s, whole job, IMBE 7200x4400.
Workload (default, BASELINE.json configs[1]): 65,536 streams per GPU x T=1 frame per step,
clean-encoded all-voiced IMBE frames, state warmed by one identical frame so both the
previous and the current model are voiced.  One "step" = one pass of the hot path
(FEC kernel + stream kernel = mbx_process_batch) over the whole batch with every input
already resident in HBM.  Streams are independent, so N GPUs = N independent shards
(weak scaling, no data-path collective); the only collective is the RCCL broadcast of the
constant-table blob at start-up (plus timing reductions).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload imbe_voiced|imbe_mixed|ambe_fec|ambe_stream]

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (codec, streams per GPU, T, description)
    "imbe_voiced": (0, 65536, 1, "BASELINE configs[1]: 65,536 IMBE 7200x4400 streams x T=1, clean all-voiced frames, warm state"),
    "imbe_mixed": (0, 65536, 16, "BASELINE configs[3]: 65,536 IMBE streams x T=16 random-bit frames (mixed voiced/unvoiced)"),
    "ambe_fec": (1, 65536, 1, "BASELINE configs[2]: 65,536 AMBE+2 streams x T=1, clean voice frames + 1% bit flips"),
    "ambe_stream": (1, 8192, 128, "BASELINE configs[4] per-GPU shard: 8,192 AMBE+2 streams x T=128 random-bit frames, int16 out"),
    # SURVEY.md §8(f) row 4 (not BASELINE configs): the other two codecs of the reference
    "imbe7100_mixed": (2, 65536, 16, "65,536 IMBE 7100x4400 streams x T=16 random-bit frames"),
    "ambe2400_mixed": (3, 65536, 16, "65,536 AMBE 3600x2400 (D-STAR) streams x T=16 random-bit frames"),
    # SURVEY.md §8(f) row 1 (not a BASELINE config): the soft-decision front end in front of the same path
    "imbe_soft": (0, 65536, 1, "65,536 IMBE 7200x4400 streams x T=1, soft-decision frames (noisy observations of random bits)"),
    "ambe_soft": (1, 65536, 1, "65,536 AMBE+2 3600x2450 streams x T=1, soft-decision frames (noisy observations of random bits)"),
}


def make_frames(name, codec, S, T, rank):
    from mbelib_neo_amd import framegen

    rng = framegen.rng_for(0xBE0000 + 97 * rank + codec)
    if name == "imbe_voiced":
        return framegen.imbe_clean_voiced_frames(S * T, rng)
    if name == "ambe_fec":
        return framegen.ambe_noisy_voice_frames(S * T, rng, ber=0.01)
    if name.endswith("_soft"):
        return framegen.soft_frames(codec, S * T, rng)
    return framegen.random_frames(codec, S * T, rng)


VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector fp32 (non-MFMA) peak


def nominal_flops_per_frame(workload):
    """SURVEY.md §8(d)'s flop model (a count of the reference's arithmetic, not of this kernel's instructions):
    voiced bank 9 flop x 160 samples x voiced components, unvoiced path 12 kflop, everything else 4 kflop.
    imbe_voiced: prev and cur components voiced on all L harmonics, mean L over uniform b0 in [0, 207] = 32.35,
    no unvoiced band; every other workload: the survey's measured random-bit mix (36 kflop of voiced bank)."""
    if workload == "imbe_voiced":
        return 9 * 160 * 2 * 32.35 + 4000, "9*160*(2*mean L = 64.7 voiced components) + 4k, no unvoiced band"
    return 36000 + 12000 + 4000, "survey's random-bit mix: 36k voiced bank + 12k unvoiced FFT path + 4k"


def algorithmic_bytes_per_launch(codec, S, T):
    """B_io = packed channel bits in + int16 PCM out per frame; B_state = load +
    store of the three-struct state per stream per launch."""
    b_io = (18 if codec in (0, 2) else 9) + 320
    return S * T * b_io + S * 2 * 3 * 2604


def measured_traffic(workload, S, T):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary of this
    workload and size (profiles/rNN/<workload>_pmc.json, written by tools/profile_round.sh: FETCH_SIZE
    and WRITE_SIZE in separate rocprofv3 passes, each calibrated on state_copy_kernel's known byte
    count).  PMC counters cannot be read from inside this process; None when no summary matches."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_pmc.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("streams_per_gpu") == S and d.get("frames_per_stream_per_step") == T and "dominant_kernel" in d:
            best = (d["dominant_kernel"]["traffic_bytes_per_launch"], os.path.relpath(path, ROOT))
    return best


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


def reference_baseline(name, codec, T, budget_s=12.0):
    """The REAL reference (oracle/_ref/libmbe_ref.so, IEEE scalar build made by oracle/Makefile from the reference's
    own sources) through oracle/tools/ref_bench.c, ONE host core, bounded sample.  None when the library is absent."""
    import ctypes as C

    path = os.path.join(ROOT, "oracle", "_ref", "libref_bench.so")
    if not os.path.exists(path) or name.endswith("_soft"):
        return None
    try:
        lib = C.CDLL(path)
    except OSError:
        return None
    from mbelib_neo_amd.layout import FRAME_CELLS, init_state

    lib.ref_process_batch.restype = C.c_int
    lib.ref_process_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    S = 4096 if T == 1 else max(64, 4096 // T)
    cells = np.ascontiguousarray(unpack_cells(codec, make_frames(name, codec, S, T, rank=0)))
    ncell = FRAME_CELLS[codec][0] * FRAME_CELLS[codec][1]
    state = np.ascontiguousarray(init_state(S))
    pcm = np.zeros((S * T, 160), dtype=np.int16)

    def once():
        rc = lib.ref_process_batch(codec, S, T, cells.ctypes.data, ncell, state.ctypes.data, 1234, pcm.ctypes.data)
        assert rc == 0, rc

    once()   # warm-up pass (also warms the state, like the GPU run)
    done, t0 = 0, time.perf_counter()
    while True:
        once()
        done += S * T
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {
        "value": done / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "reference",
        "sample": f"{done} frames ({S} streams x T={T}, same generator as the GPU workload) in {dt:.1f} s, single thread, "
                  "the reference's own sources built -O2 IEEE scalar (oracle/_ref/libmbe_ref.so), int16 output",
    }


def cpu_baseline(name, codec, T, budget_s=12.0):
    """CPU baseline on ONE host core, on a bounded sample of the same workload: the real reference when its
    library travelled with the snapshot, else the CPU oracle (a port of the reference path, oracle/mbx_oracle.c)."""
    ref = reference_baseline(name, codec, T, budget_s)
    if ref is not None:
        return ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    o = oracle_lib.load()
    soft = name.endswith("_soft")
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
        "value": done / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} frames ({S} streams x T={T}, same generator as the GPU workload) in {dt:.1f} s, single thread, "
                  "oracle/mbx_oracle.c -O2 IEEE",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="imbe_voiced", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split-expand", action="store_true", help="run the parameter expansion as a separate launch")
    ap.add_argument("--fuse-expand", action="store_true", help="development aid: IMBE at T = 1 through the fused (one-launch) path")
    ap.add_argument("--ablate", type=int, default=0, help="timing-only stage mask (mbx_debug_set_ablation); results invalid")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import mbelib_neo_amd as mbx
    from mbelib_neo_amd import _native, decoder
    from mbelib_neo_amd.parallel import broadcast_tables, shard_range

    # rank 0 reads the blob, RCCL broadcasts it, every rank uploads it and checks the checksum
    blob = broadcast_tables(mbx.load_tables_blob() if rank == 0 else None, device)
    checksum = decoder.ensure_init(local_rank, blob)

    codec, S, T, desc = WORKLOADS[args.workload]
    if args.streams:
        S = args.streams
    first, count = shard_range(S * world, world, rank)  # weak scaling: S streams on every rank
    assert count == S
    frames = make_frames(args.workload, codec, S, T, rank)
    dec = decoder.BatchDecoder(codec, S, device=local_rank, seeds=np.arange(first, first + S) + 1234, tables_blob=blob)
    d_frames = dec.to_device(frames)
    out = dec.make_outputs(T, want_pcm16=True, want_float=False, want_results=True)
    L = _native.lib()
    if args.ablate and not hasattr(L, "mbx_debug_set_ablation"):
        raise SystemExit("--ablate needs the development build: make -C mbelib-neo_amd/csrc ablate && "
                         "MBX_HIP_LIBRARY=$PWD/mbelib-neo_amd/libmbx_hip_ablate.so python bench.py --ablate MASK")
    _native.check(L.mbx_reserve(S * T), "mbx_reserve")  # launches below never allocate
    stream = torch.cuda.current_stream().cuda_stream
    soft = args.workload.endswith("_soft")
    if soft:
        def fec(frames_ptr, count, records_ptr, strm):
            return L.mbx_fec_soft(codec, frames_ptr, count, records_ptr, strm)
    else:
        fec = {0: L.mbx_fec_imbe7200x4400, 1: L.mbx_fec_ambe3600x2450, 2: L.mbx_fec_imbe7100x4400, 3: L.mbx_fec_ambe3600x2450}[codec]
    n = S * T

    def step(ev=None):
        # mbx_process_batch() = these launches (FEC stage, then mbx_process_records: the stream kernel,
        # which expands the parameter records itself); they are issued separately only so that the
        # dominant (stream) kernel can be bracketed by events
        _native.check(fec(d_frames.data_ptr(), n, out["records"].data_ptr(), stream), "fec")
        run = L.mbx_process_records
        stream_codec = 0 if codec == 2 else codec   # 7100x4400 records are in 7200x4400 order after its FEC stage
        # mbx_process_records is ONE launch for IMBE at T > 1 (expansion fused into the stream kernel); otherwise
        # it is the expand launch + the stream launch, issued separately here so that the events bracket the
        # stream kernel only.  --split-expand forces the separate launch for IMBE at T > 1 (development aid).
        split = (codec in (1, 3)) or (T == 1 and not args.fuse_expand) or args.split_expand
        if split:
            _native.check(L.mbx_expand_records(stream_codec, out["records"].data_ptr(), n, stream), "expand")
            run = L.mbx_stream_expanded
        if ev is not None:
            ev[0].record()
        _native.check(
            run(stream_codec, S, T, out["records"].data_ptr(), dec.state.data_ptr(), dec.rng.data_ptr(),
                out["pcm16"].data_ptr(), None, out["results"].data_ptr(), stream),
            "stream",
        )
        if ev is not None:
            ev[1].record()

    for _ in range(max(1, args.warmup)):  # the first pass also warms the model state
        step()
    if args.ablate:
        L.mbx_debug_set_ablation(args.ablate)  # development build only (tools/); never in a reported run
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    if world > 1:
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms = float(t[0]), float(t[1])
        cs = torch.tensor([checksum], dtype=torch.int64, device=device)
        gathered = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        assert all(int(g) == checksum for g in gathered), "table checksum differs between ranks"

    flags = decoder.results_numpy(out["results"])["flags"]
    total_frames = world * n * args.steps
    value = total_frames / dt
    alg_bytes = algorithmic_bytes_per_launch(codec, S, T)
    traffic = measured_traffic(args.workload, S, T)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    line = {
        "metric": "20ms frames/sec (whole node), " + {0: "IMBE 7200x4400", 1: "AMBE+2 3600x2450", 2: "IMBE 7100x4400", 3: "AMBE 3600x2400"}[codec],
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.ablate else f"synthetic, ABLATED stages mask={args.ablate} (timing only, INVALID as a result)",
        "config": {
            "workload": desc,
            "streams_per_gpu": S,
            "frames_per_stream_per_step": T,
            "frames_per_step": world * n,
            "output": "int16 PCM + mbe_process_result per frame",
            "parallelism": f"{world} independent stream shard(s), table blob broadcast over RCCL",
            "frame_mix": {
                "repeat": float(np.mean((flags & 0x40) != 0)),
                "mute": float(np.mean((flags & 0x80) != 0)),
                "erasure": float(np.mean((flags & 0x20) != 0)),
                "tone": float(np.mean((flags & 0x10) != 0)),
            },
        },
        "roofline": {
            "bound": "hbm",
            "kernel": L.mbx_stream_kernel_name(codec).decode(),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic[0] if traffic else None,
            "traffic_source": traffic[1] if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": kernel_ms,
            "note": "algorithmic bytes = S*T*(wire frame + int16 PCM) + S*2*3*2604 state; the path is VALU/latency bound "
                    "(SURVEY.md §8(d)), the HBM fraction is reported because BASELINE.json asks for it",
        },
    }
    fpf, fpf_basis = nominal_flops_per_frame(args.workload)
    line["valu"] = {   # "also report valu.achieved"; the resource that binds this path
        "achieved": value / world * fpf / 1e12,
        "peak": VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s per GPU",
        "frac": value / world * fpf / 1e12 / VALU_PEAK_TFLOPS,
        "flops_per_frame": fpf,
        "basis": fpf_basis,
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # timed at N = 1 only, on rank 0
            line["cpu_baseline"] = cpu_baseline(args.workload, codec, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
