#!/usr/bin/env python3
"""Per-pass timing of the forward/inverse RNS NTT at the bench shape (HIP events on the launch
stream).  PFHE_LIB_PATH selects the library build; used for tuning and ablation."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n = int(os.environ.get("LOG_N", "16"))
batch = int(os.environ.get("BATCH", "4096"))
reps = int(os.environ.get("REPS", "5"))
n, L = 1 << log_n, 3
t = p.U64DcrtTable(log_n, Q61)
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
out = []
for inverse in (0, 1):
    for i in range(p.lib().pfhe_dcrt_transform_num_passes(t._h)):
        name = p.lib().pfhe_dcrt_transform_pass_name(t._h, inverse, i).decode()
        run = lambda: check(p.lib().pfhe_dcrt_transform_pass_dev(t._h, C.c_void_p(x.data_ptr()), words, inverse, i, 0, sp))
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            run()
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append(f"{name}={ms:.3f}ms({16 * n * batch * L / ms / 1e6:.0f}GB/s)")
# whole transforms
for inverse, fn in ((0, t.transform_dev), (1, t.inverse_transform_dev)):
    fn(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn(x)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out.append(f"{'inv' if inverse else 'fwd'}_total={ms:.3f}ms({batch * L / ms / 1e3:.3f}M NTT/s)")
print(os.path.basename(os.environ.get("PFHE_LIB_PATH", "default")), f"logN={log_n} batch={batch}:", "  ".join(out))
