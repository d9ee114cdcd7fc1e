#!/bin/bash
# needs the development build of the library: make -C mbelib-neo_amd/csrc ablate; the masks do not exist in the product
export MBX_HIP_LIBRARY=${MBX_HIP_LIBRARY:-$(cd "$(dirname "$0")/.." && pwd)/mbelib-neo_amd/libmbx_hip_ablate.so}
# development aid: time the stream kernel with individual stages disabled (results invalid, timing only)
for m in ${MBX_MASKS:-0 1 2 4 8 16 32 64 128 12 28 60}; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --ablate $m ${1:+--workload $1} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mask', $m, 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'Mframes/s %.1f' % (d['value']/1e6))"
done
