"""Forward, inverse and fused NTT -> product -> INTT times at the bench shape (N = 2^16, 3 primes, batch 4096)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n, L, batch = 16, 3, 4096
n = 1 << log_n
t = p.U64DcrtTable(log_n, Q61)
x = torch.empty(batch * L * n, dtype=torch.int64, device="cuda")
b = torch.empty(L * n, dtype=torch.int64, device="cuda")
t.fill_uniform_dev(x, 1)
t.fill_uniform_dev(b, 2)
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


f = timed(lambda: t.transform_dev(x))
i = timed(lambda: t.inverse_transform_dev(x))
m = timed(lambda: t.mul_assign_dev(x, b))
pm = timed(lambda: t.mul_dcrt_polynomial_dev(x, b))
print(f"forward {f:.3f} ms, inverse {i:.3f} ms, pointwise product {m:.3f} ms, fused NTT*INTT {pm:.3f} ms "
      f"(forward + inverse = {f + i:.3f}; the fused product costs {pm - f - i:.3f} ms against {m:.3f} ms unfused)")
