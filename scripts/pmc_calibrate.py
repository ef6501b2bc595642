#!/usr/bin/env python3
"""Kernels with exactly known HBM traffic, to calibrate rocprofv3's
FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access widths (4/8/16 B per lane):
run under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`.  Buffers are
512 MiB each (> the 256 MiB Infinity Cache) so reads come from HBM."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpc_amd  # noqa: E402

lib = dpc_amd.get_library()
n = 128 * 1024 * 1024                      # floats -> 512 MiB
src = torch.rand(n, device="cuda")
dst = torch.empty(n, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for width in (1, 2, 4):
    for _ in range(3):
        lib.check(lib.dpc_debug_copy(st, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), n, width),
                  "dpc_debug_copy")
for _ in range(3):
    dst.zero_()
torch.cuda.synchronize()
print("calibration kernels done: %d bytes read and written per copy launch" % (4 * n))
