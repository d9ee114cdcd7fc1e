#!/bin/bash
# development aid: detailed SQ counters of the stream kernel (two passes of 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_IFETCH SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"; do
  rm -rf /tmp/pmc_d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_d -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${1:+--workload $1} > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob,collections,os
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_d/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if os.environ.get('MBX_KERNEL_FILTER', 'stream_kernel') in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])/float(r['Grid_Size'])*64)
print(' '.join(f"{k[3:]}={v[-1]:.0f}" for k,v in sorted(acc.items())))
PY
done
