#!/usr/bin/env python3
"""Forward / inverse RNS NTT at the bench shape: two-stream tiled transform against the pipelined single-stream form
(PFHE_PIPELINED, ntt_pipe_{fwd,inv}_kernel) for several tile counts.  Switches are read when a table is created."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
QG = [2305843009213554689, 2305843009213489153, 2305843009213317121]
log_n, batch, reps = 16, int(os.environ.get("BATCH", "4096")), int(os.environ.get("REPS", "10"))
n, L = 1 << log_n, 3
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
stream = torch.cuda.current_stream()
os.environ["PFHE_PIPELINED_MIN_MB"] = "1"
for pipe, tiles in [(0, 12)] + [(1, int(v)) for v in os.environ.get("TILES", "1,2,4,8,12,16,24,48").split(",")]:
    os.environ.pop("PFHE_DISABLE_PIPELINED", None)
    if not pipe:
        os.environ["PFHE_DISABLE_PIPELINED"] = "1"
    os.environ["PFHE_OVERLAP_TILES"] = str(max(tiles, 2))
    if tiles == 1:
        os.environ.pop("PFHE_OVERLAP_TILES")
    t = p.U64DcrtTable(log_n, Q61)
    for inverse, fn in ((0, t.transform_dev), (1, t.inverse_transform_dev)):
        for _ in range(3):
            fn(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn(x)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"pipelined={pipe} tiles={tiles:4d} {'inv' if inverse else 'fwd'} {ms:7.3f} ms  "
              f"{batch * L / ms / 1e3:.3f} M NTT/s", flush=True)
