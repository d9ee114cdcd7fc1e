#!/bin/bash
# development aid (GPU box): GRBM_GUI_ACTIVE of the stream kernel next to its duration -> effective clock
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_c
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(dict)
dur = {}
for f in glob.glob('/tmp/pmc_c/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream_kernel' in r['Kernel_Name']:
            acc[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
            if 'End_Timestamp' in r:
                dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
for f in glob.glob('/tmp/pmc_c/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream_kernel' in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
for d, c in sorted(acc.items())[-2:]:
    t = dur.get(d)
    print(d, c, "duration_s", t, "GHz", (c.get('GRBM_GUI_ACTIVE', 0) / t / 1e9) if t else None)
PY
