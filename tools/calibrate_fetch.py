#!/usr/bin/env python3
"""Calibration of FETCH_SIZE / WRITE_SIZE for the stream kernels' access pattern: state_copy_kernel
loads and stores the three structs of every stream with exactly the dword-per-lane accesses the
stream kernels use, and moves a KNOWN byte count (S * 3 * 2604 each way).  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   and   ... --pmc WRITE_SIZE ...
and compare the counter with the known bytes (MI355X_MICROARCH.md: FETCH_SIZE is uncalibrated for
access widths other than 16 B/lane)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mbelib_neo_amd import _native, decoder  # noqa: E402

S = 65536
dec = decoder.BatchDecoder(0, S)
L = _native.lib()
for _ in range(6):
    _native.check(L.mbx_state_copy(S, dec.state.data_ptr(), torch.cuda.current_stream().cuda_stream), "state_copy")
torch.cuda.synchronize()
print("known bytes per dispatch, each direction:", S * 3 * 2604)
