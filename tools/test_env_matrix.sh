#!/bin/bash
# test_env_matrix.sh -- GPU box: tests/test_gpu_parity.py under every launch-form switch of the library (each switch selects other
# kernel instances for the same calls; results must not depend on it).  About 80 s per line.  (MBX_NO_LDS_RESIDENT is not in the list: the
# full-shape tests assert the default instances BY NAME, which that switch replaces on purpose.)
cd "$(dirname "$0")/.."
for e in "MBX_FUSE_ONE=0" "MBX_FUSE_ONE=1" "MBX_SLICE_OWN=0" "MBX_SLICE=0" "MBX_SLICE_GROUPS=2" "MBX_NO_RES1=1" "MBX_FRONT_LEAD=64" "MBX_NO_REVERSE=1"; do
  echo "== $e"
  env $e python -m pytest tests/test_gpu_parity.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_default_line_keeps_its_contract 2>&1 | tail -1
done
