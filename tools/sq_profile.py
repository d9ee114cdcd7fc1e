#!/usr/bin/env python3
"""sq_profile.py -- SQ / GRBM counter evidence for the issue model of the stream kernels (GPU box).

    tools/sq_profile.py bench <workload> <out.json>     counters of the workload's dominant kernel under bench.py
    tools/sq_profile.py calib <out.json>                the same counters on tools/bin/valu_issue (known instruction streams)

Every pass is `rocprofv3 --kernel-trace --pmc <<= 8 SQ counters + GRBM_GUI_ACTIVE> -- python3 bench.py --no-extras ...`
(the program itself after `--`; --no-extras: one workload, no child processes), counters only -- never together with a
trace domain other than --kernel-trace.  This process never touches the GPU.  Counter names are filtered against
`rocprofv3 -L`, so a name this ROCm does not know drops out instead of failing the pass.

Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_BUSY_CYCLES counts per SQ (shader engine) cycles; SQ_INSTS_* count wave-instructions; GRBM_GUI_ACTIVE counts
cycles -- the calibration run pins each of them on instruction streams whose length and duration are known.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"],
    ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
     "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_FMA_F64",
     "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_INT32"],
    ["SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT64", "SQ_THREAD_CYCLES_VALU",
     "SQ_INST_CYCLES_VMEM"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_F32", "SQ_IFETCH", "SQ_INSTS_FLAT", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG", "SQ_WAIT_INST_VALU",
     "SQ_ACTIVE_INST_FLAT"],
]


def available():
    try:
        txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=300).stdout
    except Exception:   # noqa: BLE001
        return None
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "rocprof_counters.txt"), "w").write(txt)
    names = set(re.findall(r"\b((?:SQ|GRBM|TCC|TCP|TA|TD|SPI|CPC|CPF)_[A-Z0-9_]+)\b", txt))
    return names or None


def short(name):
    return name.split("(")[0].split("::")[-1].strip()


def run_pass(idx, counters, cmd):
    d = f"/tmp/sqp_{idx}"
    shutil.rmtree(d, ignore_errors=True)
    full = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--"] + cmd
    r = subprocess.run(full, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    vals = collections.defaultdict(dict)   # dispatch id -> counter -> value
    kern = {}
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            did = int(row["Dispatch_Id"])
            kern[did] = row["Kernel_Name"]
            vals[did][row["Counter_Name"]] = vals[did].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            vals[did]["_grid"] = float(row.get("Grid_Size", 0) or 0)
            vals[did]["_wg"] = float(row.get("Workgroup_Size", 0) or 0)
    dur = {}
    for f in glob.glob(d + "/*/*_kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            dur[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return vals, kern, dur, r


SIMDS, CUS, XCDS = 1024, 256, 8   # MI355X: 256 CUs x 4 SIMD-32, 8 XCDs (GRBM_GUI_ACTIVE is summed over the XCDs)
# cycles of SIMD time per wave-instruction at four waves per SIMD, measured by tools/valu_issue.hip (profiles/r03/valu_issue.jsonl):
# 8-byte VALU encodings (v_fma_f32, v_pk_*, DPP, v_fma_f64, v_mul_lo_u32) 2.9-3.3, v_mov_b32 2.07, transcendentals 6.06
COST_VALU, COST_TRANS = 3.1, 6.06
# VOP1 / VOP2 instructions with two VGPR sources and no SGPR operand retire every 2.07 cycles (v_mov, v_add_f32_e32, v_mul_f32_e32);
# the counters cannot tell them apart.  About half of the VALU instructions of the stream kernels are 4-byte encodings
# statically (an over-estimate of the cheap class: a 4-byte v_mul with an SGPR source already costs 3.03), so pricing half
# of the non-transcendental instructions at 2.07 gives a LOWER bound of the utilisation.
COST_VALU_CHEAP, CHEAP_SHARE = 2.07, 0.5


def issue_model(c, kernel_ns, frames, step_ns=None):
    """derived figures of one STEP from its counters (c: sums over the step's dispatches of the dominant kernel), the time those dispatches
    took under the profiler (kernel_ns: their durations added up -- the counter passes serialise dispatches) and the frames the step
    decoded.  step_ns: for a SLICED step (several dispatches that overlap on three queues when not profiled) the un-profiled step time;
    the utilisation figures are then taken over THAT envelope (cycles = step time x the clock the profiled dispatches ran at), not over
    the serialised sum -- instruction counts do not depend on the overlap, the time they are issued in does."""
    g = c.get
    cycles = g("GRBM_GUI_ACTIVE", 0.0) / XCDS
    if not cycles or not kernel_ns:
        return None
    clock = cycles / kernel_ns
    serialised_cycles = cycles
    if step_ns:
        cycles = step_ns * clock
    valu, trans = g("SQ_INSTS_VALU", 0.0), g("SQ_INSTS_VALU_TRANS_F32", 0.0)
    wave_q = g("SQ_WAVE_CYCLES", 0.0)
    m = {
        "clock_ghz": clock,
        "kernel_cycles": cycles,
        **({"envelope": "un-profiled step time x profiled clock (sliced step: dispatches overlap on three queues)",
            "serialised_cycles_under_pmc": serialised_cycles} if step_ns else {}),
        "waves_per_simd_resident": wave_q * 4 / (SIMDS * cycles),
        "per_frame": {k: g(n, 0.0) / frames for k, n in (("valu", "SQ_INSTS_VALU"), ("valu_trans", "SQ_INSTS_VALU_TRANS_F32"),
                                                           ("valu_f64", None), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"),
                                                           ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR"),
                                                           ("branch", "SQ_INSTS_BRANCH"), ("lds_array_cycles", "SQ_LDS_IDX_ACTIVE"),
                                                           ("lds_bank_conflict_cycles", "SQ_LDS_BANK_CONFLICT")) if n},
        # the SQ books 4 cycles per VALU instruction (8 per transcendental) whatever the SIMD needed: tools/valu_issue.hip
        # shows 4.00 / 7.96 on every stream, including ones that retire an instruction every 2.1-3.2 cycles
        "valu_busy_sq_accounting": g("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (SIMDS * cycles),
        # the same instructions priced at what the SIMD measurably needs at this occupancy
        "valu_issue_utilisation": ((valu - trans) * COST_VALU + trans * COST_TRANS) / (SIMDS * cycles),
        "valu_issue_utilisation_lower_bound": ((valu - trans) * (CHEAP_SHARE * COST_VALU_CHEAP + (1 - CHEAP_SHARE) * COST_VALU)
                                               + trans * COST_TRANS) / (SIMDS * cycles),
        "valu_cost_model": {"two_vgpr_source_vop2_cycles": COST_VALU_CHEAP, "assumed_share_of_those_for_the_lower_bound": CHEAP_SHARE,
                            "plain_or_packed_cycles": COST_VALU, "transcendental_cycles": COST_TRANS, "source": "tools/valu_issue.hip at 4 waves per SIMD"},
        "lds_array_busy": g("SQ_LDS_IDX_ACTIVE", 0.0) / (CUS * cycles),
        "lds_bank_conflict_share": (g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE", 1.0)) if g("SQ_LDS_IDX_ACTIVE") else None,
        "wave_time": ({"parked_in_waitcnt": g("SQ_WAIT_ANY", 0.0) / wave_q, "issue_stalled": g("SQ_WAIT_INST_ANY", 0.0) / wave_q,
                       "issuing": g("SQ_ACTIVE_INST_ANY", 0.0) / wave_q} if wave_q else None),
    }
    m["per_frame"]["valu_f64"] = (g("SQ_INSTS_VALU_FMA_F64", 0.0) + g("SQ_INSTS_VALU_MUL_F64", 0.0) + g("SQ_INSTS_VALU_ADD_F64", 0.0)) / frames
    m["modelled_valu_issue_utilisation"] = m["valu_issue_utilisation"]   # SQ instruction COUNTS priced with the microbenchmark's per-instruction costs: a model
    return m


def main():
    mode = sys.argv[1]
    avail = available()
    passes = [[c for c in p if avail is None or c in avail] for p in PASSES]
    dropped = [c for p in PASSES for c in p if avail is not None and c not in avail]
    out = {"mode": mode, "dropped_counters_unknown_to_this_rocm": dropped}
    if mode == "bench":
        workload, path = sys.argv[2], sys.argv[3]
        steps = 4
        cmd = ["python3", os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "2", "--min-time-ms", "0", "--no-cpu-baseline", "--no-extras", "--workload", workload]
        out["workload"] = workload
        out["command"] = " ".join(cmd)
        merged, durs = {}, []
        target, per_step, step_ns = None, 1, None
        for i, p in enumerate(passes):
            if not p:
                continue
            vals, kern, dur, r = run_pass(i, p, cmd)
            if target is None:
                try:
                    line = json.loads(r.stdout.strip().splitlines()[-1])
                    target = line["roofline"]["kernel"]
                    per_step = int(line["roofline"].get("dispatches_per_step") or 1)
                    step_ns = line["roofline"]["kernel_ms"] * 1e6 if per_step > 1 else None   # (under the profiler: only used when the line below cannot be had)
                    out["streams_per_gpu"] = line["config"]["streams_per_gpu"]
                    out["frames_per_stream_per_step"] = line["config"]["frames_per_stream_per_step"]
                except Exception:   # noqa: BLE001
                    out.setdefault("errors", []).append({"pass": i, "stdout": r.stdout[-500:], "stderr": r.stderr[-1500:]})
                    continue
            ids = sorted(d for d, k in kern.items() if short(k) == target)[-steps * per_step:]   # the timed dispatches: `steps` steps of `per_step` each
            if not ids:
                out.setdefault("errors", []).append({"pass": i, "note": "no dispatch of " + str(target), "stderr": r.stderr[-800:]})
                continue
            nsteps = len(ids) / per_step
            for c in p + ["GRBM_GUI_ACTIVE"]:
                xs = [vals[d].get(c) for d in ids if c in vals[d]]
                if xs:
                    merged.setdefault(c, []).append(sum(xs) / nsteps)   # per STEP: summed over the step's dispatches
            if all(d in dur for d in ids):
                durs.append(sum(dur[d] for d in ids) / nsteps)
            out["grid_threads"] = vals[ids[-1]].get("_grid")
        out["kernel"] = target
        out["dispatches_per_step"] = per_step
        out["counters_per_step"] = {k: sum(v) / len(v) for k, v in merged.items()}   # (= per dispatch where a step is one dispatch)
        out["kernel_ns_under_pmc"] = (sum(durs) / len(durs)) if durs else None       # the step's dispatches, durations added up
        if per_step > 1:   # the envelope of a sliced step must come from an UN-profiled run: the counter passes serialise the dispatches
            r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--min-time-ms", "300", "--no-cpu-baseline",
                                "--no-extras", "--workload", workload], capture_output=True, text=True, cwd="/tmp")
            try:
                step_ns = json.loads(r.stdout.strip().splitlines()[-1])["roofline"]["kernel_ms"] * 1e6
                out["step_ns_unprofiled"] = step_ns
            except Exception:   # noqa: BLE001
                out.setdefault("errors", []).append({"note": "no un-profiled step time", "stderr": r.stderr[-800:]})
        out["libmbx_hip_sha256_16"] = hashlib.sha256(open(os.path.join(ROOT, "mbelib-neo_amd", "libmbx_hip.so"), "rb").read()).hexdigest()[:16]
        if "streams_per_gpu" in out:
            out["issue_model"] = issue_model(out["counters_per_step"], out["kernel_ns_under_pmc"],
                                             out["streams_per_gpu"] * out["frames_per_stream_per_step"], step_ns if per_step > 1 else None)
    else:
        path = sys.argv[2]
        cmd = [os.path.join(ROOT, "tools", "bin", "valu_issue")]
        rows = {}
        text = None
        for i, p in enumerate(passes[:2] + [passes[2]]):
            if not p:
                continue
            vals, kern, dur, r = run_pass(i, p, cmd)
            text = text or r.stdout
            for did in sorted(vals):
                e = rows.setdefault(did, {"dispatch": did, "kernel": short(kern[did]), "threads": vals[did].get("_grid"), "wg": vals[did].get("_wg"), "ns": dur.get(did)})
                for c, v in vals[did].items():
                    if not c.startswith("_"):
                        e[c] = v
        out["command"] = " ".join(cmd)
        out["program_output"] = [json.loads(x) for x in (text or "").splitlines() if x.startswith("{")]
        out["dispatches"] = [rows[k] for k in sorted(rows) if "issue_kernel" in rows[k]["kernel"]]
        for e in out["dispatches"]:   # what the counters say per instruction on streams of known length
            if e.get("SQ_INSTS_VALU"):
                e["active_valu_cycles_per_valu_inst"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / e["SQ_INSTS_VALU"]
            if e.get("GRBM_GUI_ACTIVE") and e.get("ns"):
                e["waves_per_simd"] = e["wg"] / 256.0
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("dispatches", "program_output")})[:3000])


if __name__ == "__main__":
    main()
