#!/usr/bin/env python3
"""Known-byte-count streams for calibrating FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc):
calib_copy_f64 over 2^27 doubles (1 GiB read + 1 GiB written, far beyond the 256 MiB Infinity
Cache) and over 2^24 doubles (128 MiB + 128 MiB, cache-resident like the N = 2^20 env batch)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ns_gym_amd import _lib  # noqa: E402

lib = _lib.load()
for n in (1 << 27, 1 << 24):
    src = torch.ones(n, dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(10):
        _lib.check(lib.nsg_calib_copy_f64(src.data_ptr(), dst.data_ptr(), n, st))
    torch.cuda.synchronize()
    print("calib n =", n, "bytes read =", n * 8, "bytes written =", n * 8)
    del src, dst
