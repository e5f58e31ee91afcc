#!/usr/bin/env python3
"""External-product throughput at the bench shape (config 4) for tuning."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
n, L = 1 << 16, 3
batch = int(os.environ.get("BATCH", "1024"))
chunk = int(os.environ.get("CHUNK", "0"))
t = p.U64DcrtTable(16, Q61)
base = p.RNSBase(Q61)
ctx = p.DcrtGlevContext(t, base, p.BigUintApproxSignedBasis(base, 30), 1, chunk)
mods = np.array(Q61, np.uint64)


def fill(words, seed):
    x = torch.empty(words, dtype=torch.int64, device="cuda")
    check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, seed, None))
    return x


glwe, ggsw = fill(batch * 2 * L * n, 1), fill(ctx.ggsw_len(), 2)
out = torch.empty_like(glwe)
# COEFF_ONLY=1: only the coefficient-form product (counter passes: every launch then belongs to that form)
for coeff in ((True,) if os.environ.get("COEFF_ONLY") else (False, True)):
    p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=coeff)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=coeff)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"fused={'0' if os.environ.get('PFHE_DISABLE_FUSED_EXTPROD') else '1'} chunk={chunk or 'default'} batch={batch} "
          f"coeff_form={coeff}: {dt * 1e3:.2f} ms -> {batch / dt:.0f} ext-products/s ({dt / batch * 1e6:.1f} us each)")
