#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 (kernel trace or one --pmc pass at a time).

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_ntt.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/profile_ntt.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT -- python3 tools/profile_ntt.py

Runs, at the bench's shape (N=2^16, 3 primes, batch 4096 = 6 GiB, far beyond the 256 MiB Infinity
Cache): 3 forward + 1 inverse RNS NTTs, one pointwise multiply, and the external product on 64
ciphertexts.  Every kernel's algorithmic byte count is known, so FETCH_SIZE/WRITE_SIZE can be
calibrated per access pattern (MI355X_MICROARCH.md §HBM)."""
import ctypes as C
import os
import sys

# t runs the default form (large batches of N = 2^16: ntt_pipe_{fwd,inv}_kernel, tiles + 1 launches); t_plain runs one launch per
# pass over the whole batch, so that per-launch counters refer to the same launch shape as bench.py's per-kernel
# timing leg

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from primus_fhe_amd._lib import check, u64p  # noqa: E402

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
batch = int(os.environ.get("PFHE_PROFILE_BATCH", "4096"))
n, L = 1 << 16, 3
t = p.U64DcrtTable(16, Q61)
os.environ["PFHE_DISABLE_PIPELINED"] = "1"  # (read when a table is created)
t_plain = p.U64DcrtTable(16, Q61)
del os.environ["PFHE_DISABLE_PIPELINED"]
words = batch * L * n
x = torch.empty(words, dtype=torch.int64, device="cuda")
mods = np.array(Q61, np.uint64)
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 1, None))
b = x[:L * n].clone()
for _ in range(3):
    t.transform_dev(x)
t.inverse_transform_dev(x)
for _ in range(3):
    t_plain.transform_dev(x)
t.mul_assign_dev(x, b)
t_plain.inverse_transform_dev(x)
base = p.RNSBase(Q61)
ctx = p.DcrtGlevContext(t, base, p.BigUintApproxSignedBasis(base, 30), 1, 8)
ep = min(64, batch // 2)
ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int64, device="cuda")
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(ggsw.data_ptr()), ggsw.numel(), mods.ctypes.data_as(u64p), L, n, 2, None))
out = torch.empty(ep * 2 * L * n, dtype=torch.int64, device="cuda")
p.mul_dcrt_ggsw_to_dev(x[:ep * 2 * L * n], ggsw, out, ctx, into_coeff_form=True)
# fused product inside the inverse transform (config 3), shared multiplicand
t.mul_dcrt_polynomial_dev(x, b)
t_plain.mul_dcrt_polynomial_dev(x, b)
# u32 tables at the same shape (three 30-bit primes): 2 forward + 1 inverse
t32 = p.U32DcrtTable(16, [1073479681, 1071513601, 1070727169])
x32 = x.view(torch.int32)[:words]
t32.fill_uniform_dev(x32, 7)
t32.transform_dev(x32)
t32.transform_dev(x32)
t32.inverse_transform_dev(x32)
# RNS base conversion Q61 -> two 60-bit moduli over 64 Mi coefficients
cin = x[:3 * (1 << 26)]
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(cin.data_ptr()), cin.numel(), mods.ctypes.data_as(u64p), L, 1 << 26, 3, None))
conv = p.BaseConverter(base, p.RNSBase([1152921504606584833, 1152921504598720513]))
cout = torch.empty(2 * (1 << 26), dtype=torch.int64, device="cuda")
conv.fast_convert_array_dev(cin, cout, 1 << 26)
# round 6: the rows bench.py's per-leg roofline objects look up
# BASELINE config 2 (N = 2^14, one prime, batch 4096: ntt_persist_kernel), forward and inverse
t14 = p.U64DcrtTable(14, Q61[:1])
x14 = x[:4096 << 14]
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x14.data_ptr()), x14.numel(), mods.ctypes.data_as(u64p), 1, 1 << 14, 14, None))
t14.transform_dev(x14)
t14.inverse_transform_dev(x14)
# generic-prime arithmetic (Montgomery form) on the headline shape: one forward transform
check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 4, None))
os.environ["PFHE_DISABLE_PM"] = "1"
t_gen = p.U64DcrtTable(16, Q61)
del os.environ["PFHE_DISABLE_PM"]
t_gen.transform_dev(x)
# the <u32> external product on 64 ciphertexts (three 30-bit primes, log B = 15)
q30 = [1073479681, 1071513601, 1070727169]
base32 = p.RNSBase32(q30)
ctx32 = p.DcrtGlevContext32(t32, base32, p.BigUintApproxSignedBasis32(base32, 15), 1, 8)
g32 = x32[:ep * 2 * L * n]
t32.fill_uniform_dev(g32, 8)
k32 = torch.empty(ctx32.ggsw_len(), dtype=torch.int32, device="cuda")
t32.fill_uniform_dev(k32, 9)
o32 = torch.empty_like(g32)
p.mul_dcrt_ggsw_to_dev(g32, k32, o32, ctx32, into_coeff_form=True)
torch.cuda.synchronize()
print("profile workload done")
