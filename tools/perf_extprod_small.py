#!/usr/bin/env python3
"""External products at bootstrapping-sized parameters (N = 2^10..2^13, 1-2 primes) for tuning."""
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
for log_n, L, k, log_basis, batch in ((10, 1, 1, 10, 8192), (11, 1, 1, 10, 8192), (12, 2, 1, 20, 4096), (13, 2, 1, 20, 2048),
                                      (11, 1, 2, 10, 4096), (10, 1, 1, 10, 1)):
    mod = Q61[:L]
    n = 1 << log_n
    t, base = p.U64DcrtTable(log_n, mod), p.RNSBase(mod)
    basis = p.BigUintApproxSignedBasis(base, log_basis)
    ctx = p.DcrtGlevContext(t, base, basis, k)
    mods = np.array(mod, np.uint64)

    def fill(words, seed):
        x = torch.empty(words, dtype=torch.int64, device="cuda")
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, seed, None))
        return x

    glwe, ggsw = fill(batch * (k + 1) * L * n, 1), fill(ctx.ggsw_len(), 2)
    out = torch.empty_like(glwe)
    p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=True)
    torch.cuda.synchronize()
    reps = 5 if batch > 1 else 200
    t0 = time.perf_counter()
    for _ in range(reps):
        p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ell = basis.decompose_length()
    ntts = ((k + 1) * ell + (k + 1)) * L
    print(f"extprod N=2^{log_n} L={L} k={k} ell={ell} batch={batch}: {dt * 1e3:.3f} ms -> {batch / dt:.0f}/s "
          f"({dt / batch * 1e6:.2f} us each, {ntts} limb-NTTs -> {batch * ntts / dt / 1e6:.1f} M limb-NTT/s equivalent)")
