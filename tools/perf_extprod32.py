#!/usr/bin/env python3
"""The <u32> external product at the bench shape (N = 2^16, three 30-bit primes, log B = 15 -> ell = 6, k = 1, batch 1024):
fused kernels against the separate ones (PFHE_DISABLE_FUSED_EXTPROD), NTT form and coefficient form."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402

Q30 = [1073479681, 1071513601, 1070727169]
n, L = 1 << 16, 3
batch = int(os.environ.get("BATCH", "1024"))
chunk = int(os.environ.get("CHUNK", "0"))
t = p.U32DcrtTable(16, Q30)
base = p.RNSBase32(Q30)
basis = p.BigUintApproxSignedBasis32(base, int(os.environ.get("LOG_BASIS", "15")))
glwe = torch.empty(batch * 2 * L * n, dtype=torch.int32, device="cuda")
t.fill_uniform_dev(glwe, 1)
out = torch.empty_like(glwe)
# COEFF_ONLY=1 (counter passes): the fused plan's coefficient-form product only, 1 + 5 calls of `batch` products
only = bool(os.environ.get("COEFF_ONLY"))
for label, env in ((("fused", None),) if only else (("fused", None), ("separate kernels", "PFHE_DISABLE_FUSED_EXTPROD"))):
    if env:
        os.environ[env] = "1"
    try:
        ctx = p.DcrtGlevContext32(t, base, basis, 1, chunk)
    finally:
        if env:
            del os.environ[env]
    ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int32, device="cuda")
    t.fill_uniform_dev(ggsw, 2)
    for coeff in ((True,) if only else (False, True)):
        p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=coeff)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=coeff)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"u32 external product, {label}, batch={batch} chunk={chunk or 'default'} coeff_form={coeff}: {dt * 1e3:.2f} ms -> "
              f"{batch / dt:.0f} products/s ({dt / batch * 1e6:.1f} us each)")
    del ctx
