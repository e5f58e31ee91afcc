"""Stress of the host-pointer transforms: thousands of calls of random sizes on freshly allocated and on reused numpy
arrays, each compared with the device-pointer transform of the same words (bit-exact).  Run under MALLOC_CHECK_=3 to catch
heap damage at the first free.  usage: python tools/stress_host_slice.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(12345)
tables = {}
bad = 0
keep = []
for it in range(iters):
    log_n = int(rng.integers(4, 17))
    L = int(rng.integers(1, 4))
    key = (log_n, L)
    if key not in tables:
        tables[key] = p.U64DcrtTable(log_n, Q61[:L])
    t = tables[key]
    n = 1 << log_n
    max_batch = max(1, min(6, (3 << 20) // (L * n * 8)))
    batch = int(rng.integers(1, max_batch + 1))
    a = rng.integers(0, Q61[2] - 1, batch * L * n, dtype=np.uint64)
    dev = torch.from_numpy(a.view(np.int64)).cuda()
    inverse = bool(rng.integers(0, 2))
    if inverse:
        t.inverse_transform_dev(dev)
    else:
        t.transform_dev(dev)
    exp = dev.cpu().numpy().view(np.uint64)
    x = a.copy() if rng.integers(0, 2) else a
    (t.inverse_transform_slice if inverse else t.transform_slice)(x)
    if not np.array_equal(x, exp):
        bad += 1
        print(f"MISMATCH it={it} logN={log_n} L={L} batch={batch} inverse={inverse} "
              f"first bad word {int(np.flatnonzero(x != exp)[0])} of {x.size}")
    if rng.integers(0, 4) == 0:
        keep.append(x)       # keep some arrays alive so that the heap layout keeps changing
    if len(keep) > 8:
        keep.pop(int(rng.integers(0, len(keep))))
print(f"{iters} calls, {bad} mismatches, alloc events {p.lib().pfhe_debug_alloc_count()}")
sys.exit(1 if bad else 0)
