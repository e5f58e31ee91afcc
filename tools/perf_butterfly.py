"""Streaming rate of the GLWE butterfly (a, b) = (a + s, (a - s) * w) on 3 GiB operands (N = 2^16, 3 limbs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n, L, polys = 16, 3, 2048
n = 1 << log_n
t = p.U64DcrtTable(log_n, Q61)
words = polys * L * n
a, s, w, b = (torch.empty(words, dtype=torch.int64, device="cuda") for _ in range(4))
for i, x in enumerate((a, s, w)):
    t.fill_uniform_dev(x, i + 1)
f = torch.empty(2 * L * n, dtype=torch.int64, device="cuda")
t.fill_uniform_dev(f, 9)
st = torch.cuda.current_stream()


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn, nbytes in (("butterfly_mul_dcrt_polynomial_to, shared w", lambda: t.butterfly_mul_dcrt_polynomial_to_dev(a, s, w[:L * n], b), 32 * words),
                         ("butterfly_mul_dcrt_polynomial_to, per-element w", lambda: t.butterfly_mul_dcrt_polynomial_to_dev(a, s, w, b), 40 * words),
                         ("butterfly_mul_factor_to, shared factors", lambda: t.butterfly_mul_factor_to_dev(a, s, f, b), 32 * words)):
    ms = timed(fn)
    print(f"{name:50s} {ms:8.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:.0f} % of 8 TB/s)")
