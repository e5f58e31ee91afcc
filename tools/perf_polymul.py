"""NTT -> product -> INTT at the bench shape (N = 2^LOG_N [16], 3 primes, 6 GiB batch), per tuning environment.
Usage: python tools/perf_polymul.py [ENV=VAL,ENV=VAL ...]   (one table per argument; switches are read at table creation)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
log_n = int(os.environ.get("LOG_N", "16"))
L, batch = 3, int(os.environ.get("BATCH", str(4096 << (16 - log_n))))
n = 1 << log_n
x = torch.empty(batch * L * n, dtype=torch.int64, device="cuda")
b = torch.empty(L * n, dtype=torch.int64, device="cuda")
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


for spec in (sys.argv[1:] or [""]):
    env = dict(kv.split("=", 1) for kv in spec.split(",") if kv)
    os.environ.update(env)
    try:
        t = p.U64DcrtTable(log_n, Q61)
    finally:
        for k in env:
            del os.environ[k]
    t.fill_uniform_dev(x, 1)
    t.fill_uniform_dev(b, 2)
    pm = timed(lambda: t.mul_dcrt_polynomial_dev(x, b))
    f = timed(lambda: t.transform_dev(x))
    i = timed(lambda: t.inverse_transform_dev(x))
    print(f"{spec or 'default':40s} polymul {pm:.3f} ms = {batch / pm:.1f} k/s   forward {f:.3f}  inverse {i:.3f}", flush=True)
