#!/bin/bash
# development aid (GPU box): headline numbers of every bench workload
for w in ${@:-imbe_voiced imbe_mixed ambe_fec ambe_stream}; do
  timeout 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload $w 2>&1 | tail -1 > /tmp/line.json
  python3 - "$w" <<'PY'
import sys, json
d = json.loads(open('/tmp/line.json').read())
print(sys.argv[1], "frames/s %.4g" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
done
