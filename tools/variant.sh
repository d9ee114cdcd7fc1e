#!/bin/bash
# variant.sh <name> <extra hipcc flags...> -- build a variant of the library into mbelib-neo_amd/variants/libmbx_hip_<name>.so
# (development aid for A/B timing on the GPU box: MBX_HIP_LIBRARY=... python bench.py ...)
cd "$(dirname "$0")/../mbelib-neo_amd/csrc" || exit 1
name=$1; shift
mkdir -p ../variants
make -s EXTRA="$*" OUT=$(pwd)/../variants/libmbx_hip_$name.so $(pwd)/../variants/libmbx_hip_$name.so
