#!/bin/bash
# development aid (GPU box): rebuild the stream kernels with extra -D flags and time workloads
# usage: tools/exp_waves.sh "<workloads>" "<EXTRA flags 1>" "<EXTRA flags 2>" ...
WL=$1; shift
for x in "$@"; do
  make -C mbelib-neo_amd/csrc -B EXTRA="$x" > /dev/null 2>&1
  echo "== EXTRA=$x"
  tools/bench_all.sh $WL
done
