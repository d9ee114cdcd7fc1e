#!/bin/bash
# Collects the judged evidence for one bench workload on the GPU box:
#   FETCH_SIZE and WRITE_SIZE PMC passes (separate passes), plus the same two counters
#   on state_copy_kernel, which moves a KNOWN byte count with the stream kernels' dword-per-lane accesses (the calibration
#   MI355X_MICROARCH.md asks for before trusting an absolute FETCH_SIZE at an access width other than 16 B/lane), the SQ
#   counter passes of tools/sq_profile.py, rocprofv3 --kernel-trace --stats, and LAST the un-profiled bench line -- after the two summaries have been
#   copied to profiles/<round>/ on the box, so that the line carries the traffic and issue figures of this very build.
# usage: tools/profile_round.sh <out-prefix> [workload]   -> gpurun_out/<out-prefix>_{bench.json,kernel_stats.csv,pmc.json,sq.json}
#        (out-prefix = <round>_<workload>, e.g. r03_imbe_voiced: the summaries land in profiles/r03/imbe_voiced_*.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
WL=${2:-imbe_voiced}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROUND=${TAG%%_*}
python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --workload $WL > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err   # kernel name for the summary
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/c_fetch /tmp/c_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --min-time-ms 0 --no-cpu-baseline --no-extras --workload $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --steps 5 --warmup 1 --min-time-ms 0 --no-cpu-baseline --no-extras --workload $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/c_fetch -- python3 $R/tools/calibrate_fetch.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/c_write -- python3 $R/tools/calibrate_fetch.py > /dev/null 2>&1
python3 - "$OUT/${TAG}_pmc.json" "$OUT/${TAG}_bench.json" "$WL" <<'PY'
import csv, glob, collections, hashlib, json, os, sys

def mean_kb(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

known = 65536 * 3 * 2604  # bytes each way per state_copy_kernel dispatch (tools/calibrate_fetch.py)
fetch, write = mean_kb("/tmp/p_fetch"), mean_kb("/tmp/p_write")
cf, cw = mean_kb("/tmp/c_fetch"), mean_kb("/tmp/c_write")
k_copy = [k for k in cf if "state_copy" in k][0]
f_fac = known / (cf[k_copy][0] * 1024.0)
w_fac = known / (cw[k_copy][0] * 1024.0)
bench = json.load(open(sys.argv[2]))
kern = bench["roofline"]["kernel"]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {
    "workload": sys.argv[3],
    # bench.py reports this file's traffic only next to timings of the SAME build of the library
    "libmbx_hip_sha256_16": hashlib.sha256(open(os.path.join(root, "mbelib-neo_amd", "libmbx_hip.so"), "rb").read()).hexdigest()[:16],
    "streams_per_gpu": bench["config"]["streams_per_gpu"],
    "frames_per_stream_per_step": bench["config"]["frames_per_stream_per_step"],
    "unit": "KB per dispatch (rocprofv3 Counter_Value), mean over dispatches",
    "counters": [{"counter": n, "kernel": k, "dispatches": v[1], "mean_value_KB": v[0]}
                 for n, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)) for k, v in d.items()],
    "calibration": {
        "kernel": k_copy, "known_bytes_each_way": known,
        "FETCH_SIZE_KB": cf[k_copy][0], "WRITE_SIZE_KB": cw[k_copy][0],
        "fetch_factor": f_fac, "write_factor": w_fac,
        "note": "factor = known bytes / counter bytes for dword-per-lane coalesced accesses (the stream kernels' pattern)",
    },
}
fk = [k for k in fetch if k.split("::")[-1] == kern] or [k for k in fetch if kern in k]
if fk:
    k = fk[0]
    # a step = ONE mbx_process_batch call = `n` dispatches of the dominant kernel (1, or the 3 x ceil(T / slice) slices of a sliced launch,
    # every one of them a dispatch of the same kernel: the step's bytes are the mean per dispatch x n)
    n = int(bench["roofline"].get("dispatches_per_step") or 1)
    rd, wr = fetch[k][0] * 1024.0 * f_fac, write[k][0] * 1024.0 * w_fac
    out["dominant_kernel"] = {"kernel": k, "dispatches_per_step": n,
                              "read_bytes_per_dispatch": rd, "write_bytes_per_dispatch": wr,
                              "read_bytes_per_launch": rd * n, "write_bytes_per_launch": wr * n,
                              "traffic_bytes_per_launch": (rd + wr) * n,
                              "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
                              "traffic_over_algorithmic": (rd + wr) * n / bench["roofline"]["algorithmic_bytes_per_launch"]}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out.get("dominant_kernel")), json.dumps(out["calibration"]))
PY
python3 $R/tools/sq_profile.py bench $WL $OUT/${TAG}_sq.json > $OUT/${TAG}_sq.log 2>&1
mkdir -p $R/profiles/$ROUND
cp $OUT/${TAG}_pmc.json $R/profiles/$ROUND/${WL}_pmc.json
cp $OUT/${TAG}_sq.json $R/profiles/$ROUND/${WL}_sq.json
# the --stats pass directly in front of the un-profiled line: the two are compared (profiles/README.md), and a box that has been under load for
# minutes runs the same kernel up to 9 % slower than a fresh one
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --workload $WL > /dev/null 2>&1
cp /tmp/p_stats/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 $R/bench.py --steps 50 --warmup 5 --workload $WL > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cp $R/bench_detail.json $OUT/${TAG}_bench_detail.json   # the full measurement behind the compact line
cut -c1-60,150-230 $OUT/${TAG}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench.json')); print(d['value'], d['roofline'], d['cpu_baseline'])"
