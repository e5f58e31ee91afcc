"""Streaming rate of the element-wise family at BASELINE config 4's data size (3 GiB of CrtGlwe: N = 2^16,
3 limbs, k = 1, 1024 ciphertexts).  Bytes = compulsory traffic (each operand read once, result written once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n = int(os.environ.get("LOGN", "16"))
n, L = 1 << log_n, 3
polys = int(os.environ.get("POLYS", str(2048 << (16 - log_n))))
t = p.U64DcrtTable(log_n, Q61)
words = polys * L * n
a, b, o = (torch.empty(words, dtype=torch.int64, device="cuda") for _ in range(3))
t.fill_uniform_dev(a, 1)
t.fill_uniform_dev(b, 2)
a.clamp_(min=1)
scalars = [q - 2 for q in Q61]
factors = [v for s, q in zip(scalars, Q61) for v in (s, (s << 64) // q)]
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


print(f"N=2^{log_n} L={L} polynomials={polys} ({words * 8 / 2**30:.2f} GiB per operand)")
for name, fn, nbytes in (
        ("add_to", lambda: t.add_to_dev(a, b, o), 24 * words),
        ("add_assign", lambda: t.add_to_dev(o, b, o), 24 * words),
        ("sub_to", lambda: t.sub_to_dev(a, b, o), 24 * words),
        ("neg_to", lambda: t.neg_to_dev(a, o), 16 * words),
        ("mul_scalar_to", lambda: t.mul_scalar_to_dev(a, scalars, o), 16 * words),
        ("mul_factor_to", lambda: t.mul_factor_to_dev(a, factors, o), 16 * words),
        ("add_mul_scalar_assign", lambda: t.add_mul_scalar_assign_dev(o, b, scalars), 24 * words),
        ("add_mul_factor_assign", lambda: t.add_mul_factor_assign_dev(o, b, factors), 24 * words),
        ("mul_monomial_to r=12345", lambda: t.mul_monomial_to_dev(a, 12345 % (2 * n), o), 16 * words),
        ("mul_monomial_to r=N+2", lambda: t.mul_monomial_to_dev(a, n + 2, o), 16 * words),
        ("mul_monomial_assign (in place)", lambda: t.mul_monomial_assign_dev(o, 12345 % (2 * n)), 16 * words),
        ("inv_to", lambda: t.inv_to_dev(a, o), 16 * words)):
    ms = timed(fn, 3 if name == "inv_to" else 10)
    print(f"{name:26s} {ms:8.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:.0f} % of 8 TB/s)")
