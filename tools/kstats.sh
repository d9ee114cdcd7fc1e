#!/bin/bash
# development aid (GPU box): per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ks/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print("%-28s calls %4s avg %9.1f us  %5s %%" % (r['Name'].split('(')[0][-28:], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
