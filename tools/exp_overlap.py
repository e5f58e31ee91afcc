#!/usr/bin/env python3
"""Experiment: overlap the memory-bound strided pass of tile k+1 with the VALU-bound block pass of
tile k on two HIP streams (forward NTT, bench shape)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
n, L, batch = 1 << 16, 3, 4096
t = p.U64DcrtTable(16, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(tiles):
    tw = words // tiles
    evs = []
    for k in range(tiles):
        ptr = C.c_void_p(x.data_ptr() + 8 * tw * k)
        check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, ptr, tw, 0, 0, 0, C.c_void_p(sa.cuda_stream)))
        e = torch.cuda.Event()
        e.record(sa)
        sb.wait_event(e)
        check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, ptr, tw, 0, 1, 0, C.c_void_p(sb.cuda_stream)))


for tiles in (1, 2, 4, 8, 16, 32, 64):
    run(tiles)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        run(tiles)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"tiles={tiles:3d}: {ms:.3f} ms per forward NTT of the batch -> {batch * L / ms / 1e3:.3f} M NTT/s")
