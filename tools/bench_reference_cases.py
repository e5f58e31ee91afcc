#!/usr/bin/env python3
"""The reference's own criterion cases (crates/primus_ntt/benches/bench_u64.rs:8,117-130 and
bench_u32.rs:8: one N = 4096 transform per iteration) on this host's CPU (oracle restatement, scalar
and AVX-512) and on the GPU (one polynomial per call through the device API, and a resident batch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import primus_fhe_amd as p
from oracle import oracle

LOG_N, N = 12, 4096


def cpu_time(fn, x, reps=2000):
    fn(x)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(x)
    return (time.perf_counter() - t0) / reps * 1e6


def gpu_time(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


rows = []
for q in (1073692673, 1125899906826241):
    o, t = oracle.U64NttTable(LOG_N, q), p.U64NttTable(LOG_N, q)
    a = np.random.default_rng(0).integers(0, q, N, dtype=np.uint64)
    cpu_s = cpu_time(o.transform_slice, a.copy())
    cpu_v = cpu_time(o.transform_slice_avx512, a.copy()) if oracle.lib().orc_avx512_available() else float("nan")
    d1 = torch.from_numpy(a.view(np.int64)).cuda()
    g1 = gpu_time(lambda: t.transform_dev(d1))
    batch = 65536
    db = torch.from_numpy(np.tile(a, batch).view(np.int64)).cuda()
    gb = gpu_time(lambda: t.transform_dev(db), 20) / batch
    rows.append((f"U64NttTable FWD q={q} N=4096", cpu_s, cpu_v, g1, gb))
q32 = 268369921
o32, t32 = oracle.U32NttTable(LOG_N, q32), p.U32NttTable(LOG_N, q32)
a32 = np.random.default_rng(1).integers(0, q32, N, dtype=np.uint64).astype(np.uint32)
d1 = torch.from_numpy(a32.view(np.int32)).cuda()
db = torch.from_numpy(np.tile(a32, 65536).view(np.int32)).cuda()
rows.append((f"U32NttTable FWD q={q32} N=4096", cpu_time(o32.transform_slice, a32.copy()), float("nan"),
             gpu_time(lambda: t32.transform_dev(d1)), gpu_time(lambda: t32.transform_dev(db), 20) / 65536))
print(f"{'case':48s} {'CPU scalar us':>14s} {'CPU AVX-512 us':>15s} {'GPU 1 poly us':>14s} {'GPU batch us/poly':>18s}")
for r in rows:
    print(f"{r[0]:48s} {r[1]:14.2f} {r[2]:15.2f} {r[3]:14.2f} {r[4]:18.4f}")
