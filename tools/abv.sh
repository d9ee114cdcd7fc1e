#!/bin/bash
# abv.sh <workload> <variant names...> -- GPU box: bench line of the product library and of each variant library
w=$1; shift
cd "$(dirname "$0")/.."
run() {  # label, env...
  label=$1; shift
  env "$@" python bench.py --workload $w --steps 10 --no-cpu-baseline --no-extras > gpurun_out/abv_$label.log 2>&1
  python - "$label" <<'PY'
import json, sys
f = "gpurun_out/abv_%s.log" % sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"] / 1e6, 1), "M frames/s, kernel", round(d["roofline"]["kernel_ms"], 4), "ms")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(f).read()[-300:])
PY
}
run product X=1
for v in "$@"; do run $v MBX_HIP_LIBRARY=$PWD/mbelib-neo_amd/variants/libmbx_hip_$v.so; done
