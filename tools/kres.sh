#!/bin/bash
# kres.sh [file.hip] [extra hipcc flags...] -- compact per-kernel resource table (VGPRs, SGPRs, scratch, occupancy, LDS)
# of one csrc/ source, compiled with the product flags.  Runs without a GPU.
cd "$(dirname "$0")/../mbelib-neo_amd/csrc" || exit 1
src=${1:-mbx_stream.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
  -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /tmp/kres_$$.o 2>&1 |
awk '/Function Name:/ {name=$(NF-1); sub(/^_ZN3mbx[0-9]+/,"",name); sub(/E[iPN].*$/,"",name)}
     /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)}
     /LDS Size/ {printf "%-34s vgpr %3s sgpr %3s scratch %4s occ %2s lds %6s\n", name, v, s, sc, o, $(NF-1)}'
rm -f /tmp/kres_$$.o
