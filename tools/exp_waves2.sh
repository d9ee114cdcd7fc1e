#!/bin/bash
# development aid (GPU box): like exp_waves.sh, with extra bench arguments per run
# usage: tools/exp_waves2.sh "<workload> [bench args]" "<EXTRA flags 1>" ...
WL=$1; shift
for x in "$@"; do
  make -C mbelib-neo_amd/csrc -B EXTRA="$x" > /dev/null 2>&1
  echo "== EXTRA=$x"
  python3 bench.py --no-cpu-baseline --workload $WL | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' frames/s %.4g kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms']))"
done
