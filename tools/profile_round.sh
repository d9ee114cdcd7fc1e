#!/bin/bash
# Collects the judged evidence for one bench workload on the GPU box:
#   bench line (un-profiled), rocprofv3 --stats, FETCH_SIZE and WRITE_SIZE PMC passes.
# usage: tools/profile_round.sh <tag> [workload]      -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
WL=${2:-imbe_voiced}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 50 --warmup 5 --workload $WL > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL > /dev/null 2>&1
cp /tmp/p_stats/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 - "$OUT/${TAG}_pmc.json" <<'PY'
import csv,glob,collections,json,sys
out=[]
for name,d in (("FETCH_SIZE","/tmp/p_fetch"),("WRITE_SIZE","/tmp/p_write")):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k,v in acc.items():
        out.append({"counter":name,"kernel":k,"dispatches":len(v),"mean_value_KB":sum(v)/len(v)})
json.dump(out,open(sys.argv[1],"w"),indent=1)
PY
cat $OUT/${TAG}_kernel_stats.csv | cut -c1-50,120-220
cat $OUT/${TAG}_pmc.json | tr -d '\n' | cut -c1-900; echo
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench.json')); print(d['value'], d['roofline'], d['cpu_baseline'])"
