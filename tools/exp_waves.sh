#!/bin/bash
# development aid (GPU box): rebuild the stream kernels for several occupancy targets and time them
for w in "$@"; do
  make -C mbelib-neo_amd/csrc -B EXTRA=-DMBX_STREAM_WAVES_PER_SIMD=$w > /dev/null 2>&1
  echo "== waves/SIMD target $w"
  tools/bench_all.sh imbe_voiced imbe_mixed
done
