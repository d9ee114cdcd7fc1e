#!/bin/bash
# quick4.sh [label] -- GPU box: kernel time of the four BASELINE workloads with the library in the tree (development aid)
cd "$(dirname "$0")/.."
L=${1:-q}
for w in imbe_voiced imbe_voiced_resident imbe_mixed ambe_fec ambe_stream; do
  python bench.py --workload $w --steps 10 --no-cpu-baseline --no-extras > gpurun_out/${L}_$w.log 2> gpurun_out/${L}_$w.err || { tail -5 gpurun_out/${L}_$w.err; exit 1; }
  python - "$w" "gpurun_out/${L}_$w.log" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"] / 1e6, 1), "M/s", d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), "ms  step", round(d["ms_per_step"], 4))
PY
done
