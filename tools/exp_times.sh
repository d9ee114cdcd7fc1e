#!/bin/bash
# development aid (GPU box): stream-kernel time for ablation masks / stream counts (arguments passed to bench.py per line)
run() {
  timeout 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline $1 2>&1 | tail -1 > /tmp/line.json
  python3 - "$1" <<'PY'
import sys, json
d = json.loads(open('/tmp/line.json').read())
print(sys.argv[1], "| frames/s %.4g" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "ms_per_step %.4f" % d["ms_per_step"])
PY
}
for a in "$@"; do run "$a"; done
