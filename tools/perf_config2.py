#!/usr/bin/env python3
"""BASELINE config 2: N = 2^14, one 61-bit prime, batch 4096 forward / inverse NTT (single block pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch

import primus_fhe_amd as p
from primus_fhe_amd._lib import check, u64p

q = int(os.environ.get("Q", "2305843009211596801"))
log_n = int(os.environ.get("LOG_N", "14"))
batch = int(os.environ.get("BATCH", "4096"))
n = 1 << log_n
t = p.U64NttTable(log_n, q)
x = torch.empty(batch * n, dtype=torch.int64, device="cuda")
mods = np.array([q], np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), x.numel(), mods.ctypes.data_as(u64p), 1, n, 2, None))
stream = torch.cuda.current_stream()
for name, fn in (("forward", t.transform_dev), ("inverse", t.inverse_transform_dev)):
    for _ in range(int(os.environ.get("WARM", "600"))):  # ~0.25 s: the shader clock needs that long to settle
        fn(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("REPS", "200"))
    e0.record(stream)
    for _ in range(reps):
        fn(x)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"config2 logN={log_n} batch={batch} {name}: {ms:.3f} ms -> {batch / ms / 1e3:.2f} M NTT/s, "
          f"{16 * n * batch / ms / 1e6:.0f} GB/s = {16 * n * batch / ms / 1e6 / 80:.1f} % of 8 TB/s")
