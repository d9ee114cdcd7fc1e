#!/bin/bash
# mkbase.sh [rev=HEAD] [name=base] -- build libmbx_hip of a committed revision into mbelib-neo_amd/variants/libmbx_hip_<name>.so
# (the A side of tools/abx.sh; development aid)
rev=${1:-HEAD}; name=${2:-base}
root="$(cd "$(dirname "$0")/.." && pwd)"
rm -rf /tmp/mkbase_tree && mkdir -p /tmp/mkbase_tree "$root/mbelib-neo_amd/variants"
git -C "$root" archive "$rev" mbelib-neo_amd/csrc include | tar -x -C /tmp/mkbase_tree
out="$root/mbelib-neo_amd/variants/libmbx_hip_$name.so"
make -s -C /tmp/mkbase_tree/mbelib-neo_amd/csrc OUT="$out" "$out" 2>&1 | grep -E "error" ; ls -la "$out"
