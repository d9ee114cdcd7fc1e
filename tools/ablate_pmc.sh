#!/bin/bash
# needs the development build of the library: make -C mbelib-neo_amd/csrc ablate; the masks do not exist in the product
export MBX_HIP_LIBRARY=${MBX_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/mbelib-neo_amd/libmbx_hip_ablate.so}
# development aid: VALU/SALU/LDS instruction counts of the stream kernel per ablation mask
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in ${MBX_MASKS:-0 1 2 4 8 16 32 64 128}; do
  rm -rf /tmp/pmc_abl
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_abl -- python3 $R/bench.py --steps 2 --warmup 1 --min-time-ms 0 --no-cpu-baseline --no-extras --ablate $m ${1:+--workload $1} $MBX_BENCH_ARGS > /dev/null 2>&1
  python3 - "$m" <<'PY'
import csv,glob,os,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_abl/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].split('(')[0].split('::')[-1].strip() == os.environ.get('MBX_KERNEL', 'imbe_stream_kernel'):   # exact name (MBX_KERNEL=...)
            acc[r['Counter_Name']].append(float(r['Counter_Value'])/float(r['Grid_Size'])*64)
# the last dispatches are the ablated ones (warmup runs un-ablated)
print('mask',sys.argv[1],' '.join(f"{k[3:]}={v[-1]:.0f}" for k,v in sorted(acc.items())))
PY
done
