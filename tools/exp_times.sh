#!/bin/bash
# development aid (GPU box): stream-kernel time for stream counts / ablation masks
run() {
  timeout 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/line.json
  python3 - "$*" <<'PY'
import sys, json
d = json.loads(open('/tmp/line.json').read())
print(sys.argv[1], "| frames/s %.4g" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "ms_per_step %.4f" % d["ms_per_step"])
PY
}
for s in 16384 32768 65536 131072 262144; do run --streams $s; done
for m in 128 132 255; do run --ablate $m; done
