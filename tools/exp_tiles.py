#!/usr/bin/env python3
"""Forward RNS NTT at the bench shape for several tile counts of the two-stream transform (switches are read when
a table is created, so every setting gets a table of its own).  Question: with tiles small enough to stay in the 256 MiB Infinity Cache, does the block pass read its
input on-die (one HBM read + one HBM write per transform instead of two of each)?"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n, batch, reps = 16, int(os.environ.get("BATCH", "4096")), int(os.environ.get("REPS", "10"))
n, L = 1 << log_n, 3
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
stream = torch.cuda.current_stream()
for tiles in [int(v) for v in os.environ.get("TILES", "1,12,16,24,32,48,64,96,128,192,256").split(",")]:
    if tiles == 1:
        os.environ["PFHE_DISABLE_OVERLAP"] = "1"
    else:
        os.environ.pop("PFHE_DISABLE_OVERLAP", None)
        os.environ["PFHE_OVERLAP_TILES"] = str(tiles)
    os.environ["PFHE_OVERLAP_INVERSE"] = "1"
    t = p.U64DcrtTable(log_n, Q61)
    for inverse, fn in ((0, t.transform_dev), (1, t.inverse_transform_dev)):
        fn(x)
        fn(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn(x)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"tiles={tiles:4d} tile={words * 8 / max(tiles, 1) / 2**20:8.1f} MiB {'inv' if inverse else 'fwd'} "
              f"{ms:7.3f} ms  {batch * L / ms / 1e3:.3f} M NTT/s", flush=True)
