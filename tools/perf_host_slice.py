"""Latency / rate of the host-pointer (`*_slice`) entry points: pageable numpy arrays in, the same arrays out.
Reports per-call time, bytes of the slice / time (GB/s, one direction counted, as DESIGN.md quotes it) and the number
of allocation events across the timed calls (pfhe_debug_alloc_count: must be 0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import primus_fhe_amd as p

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
print("PFHE_STAGE_CHUNK =", os.environ.get("PFHE_STAGE_CHUNK"), " PFHE_STAGE_BOUNCE_MAX =", os.environ.get("PFHE_STAGE_BOUNCE_MAX"))
for log_n, L, batch in ((12, 1, 1), (16, 1, 1), (16, 3, 1), (16, 3, 16), (16, 3, 64)):
    t = p.U64DcrtTable(log_n, Q61[:L])
    a = np.random.default_rng(0).integers(0, Q61[0] - 10**6, batch * L << log_n, dtype=np.uint64)
    for _ in range(3):
        t.transform_slice(a)
    reps = 50 if a.nbytes < (8 << 20) else 10
    c0 = p.lib().pfhe_debug_alloc_count()
    best, t_all = 1e9, time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        t.transform_slice(a)
        best = min(best, time.perf_counter() - t0)
    dt = (time.perf_counter() - t_all) / reps
    c1 = p.lib().pfhe_debug_alloc_count()
    print(f"transform_slice logN={log_n} L={L} batch={batch}: {dt*1e6:.0f} us per call (best {best*1e6:.0f}), "
          f"{a.nbytes/dt/1e9:.2f} GB/s, alloc events {c1-c0}")
