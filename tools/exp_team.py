#!/usr/bin/env python3
"""EXPERIMENT: forward transform as one persistent launch with the strided -> block hand-off through the XCD's L2
(PFHE_TEAM=1, ntt_team_fwd_kernel) against the default pipelined form: bit-exact comparison of the whole batch, then
timings for several lags / workgroups per CU."""
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
mods = np.array(Q61, np.uint64)
x = torch.empty(words, dtype=torch.int64, device="cuda")
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
orig = x.clone()
ref_t = p.U64DcrtTable(log_n, Q61)
ref = orig.clone()
ref_t.transform_dev(ref)
stream = torch.cuda.current_stream()


def run(t, label):
    y = orig.clone()
    t.transform_dev(y)
    torch.cuda.synchronize()
    ok = bool(torch.equal(y, ref))
    for _ in range(2):
        t.transform_dev(y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        t.transform_dev(y)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:28s} bit-exact={ok}  {ms:7.3f} ms  {batch * L / ms / 1e3:.3f} M NTT/s", flush=True)


run(ref_t, "default (pipelined)")
os.environ["PFHE_TEAM"] = "1"
for wgs in os.environ.get("WGS", "4,3,2").split(","):
    for lag in os.environ.get("LAGS", "2,3,4,6").split(","):
        os.environ["PFHE_TEAM_LAG"], os.environ["PFHE_TEAM_WGS"] = lag, wgs
        run(p.U64DcrtTable(log_n, Q61), f"team lag={lag} wgs/CU={wgs}")
