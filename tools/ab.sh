#!/bin/bash
# ab.sh <workload> [env assignment for the B run] -- GPU box: one bench line per variant (value M frames/s, kernel, kernel ms)
w=${1:-imbe_mixed}; shift
cd "$(dirname "$0")/.."
python bench.py --workload $w --steps 10 --no-cpu-baseline --no-extras > gpurun_out/ab_a.log 2>&1
env "$@" python bench.py --workload $w --steps 10 --no-cpu-baseline --no-extras > gpurun_out/ab_b.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/ab_a.log", "gpurun_out/ab_b.log"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"] / 1e6, 1), d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4))
    except Exception as e:
        print(f, "FAILED", e, open(f).read()[-400:])
PY
