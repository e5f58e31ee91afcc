"""Stress of the host-pointer transforms (`transform_slice` / `inverse_transform_slice`, primus_ntt/src/ntt/prime64/
table.rs:541-563: `&self, &mut [T]`, any memory).  Thousands of calls of random sizes on freshly allocated and on reused
numpy arrays, every result compared with the ORACLE's transform of the same words (expected outputs are computed once per
source polynomial, a call's input is a random concatenation of sources).

    python tools/stress_host_slice.py [iterations] [--register] [--churn-threads N] [--seed S]

--register        the CALLER pins every slice for the call: hipHostRegister -> transform -> hipHostUnregister.  The library
                  then takes its caller-pinned path: kernels read and write the mapped range themselves (slices up to
                  PFHE_STAGE_BOUNCE_MAX) or the copy engines move it.  This is the configuration round 4's library-side
                  registration produced wrong words in (profiles/r04_experiments.txt, item 6), driven through the public ABI.
--churn-threads N N host threads keep the C heap in motion meanwhile (tests/native/heap_churn.c: arenas growing and
                  trimming, mmap-sized blocks coming and going at reused addresses), and report blocks whose guard bytes
                  were damaged.
Run under MALLOC_CHECK_=3 as well (keeps large arrays on the heap, so addresses are reused from call to call; aborts at the
first damaged heap header).  Exit status 0 = every result bit-exact and no damaged block."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import primus_fhe_amd as p  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker)

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
ap = argparse.ArgumentParser()
ap.add_argument("iters", nargs="?", type=int, default=5000)
ap.add_argument("--register", action="store_true")
ap.add_argument("--churn-threads", type=int, default=0)
ap.add_argument("--seed", type=int, default=12345)
ap.add_argument("--callers", type=int, default=1, help="caller threads driving the entry points at the same time (one table "
                "handle per shape, shared: NttTable is Send + Sync)")
ap.add_argument("--max-bytes", type=int, default=3 << 20, help="largest slice (above PFHE_STAGE_BOUNCE_MAX = 1 MiB a slice "
                "takes the copy engines instead of the zero-copy kernels)")
args = ap.parse_args()

rng = np.random.default_rng(args.seed)
rt = torch.cuda.cudart()
torch.cuda.init()

# ---- heap churn threads (C, the GIL is released while they run)
stop = C.c_int(0)
churners, damaged = [], []
if args.churn_threads:
    so = os.path.join(tempfile.mkdtemp(prefix="pfhe_churn_"), "libheap_churn.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "heap_churn.c")], check=True)
    lib = C.CDLL(so)
    lib.heap_churn.restype = C.c_uint64
    lib.heap_churn.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
    for k in range(args.churn_threads):
        th = threading.Thread(target=lambda k=k: damaged.append(int(lib.heap_churn(args.seed * 1000 + k, 1 << 62, C.byref(stop)))))
        th.start()
        churners.append(th)

# ---- sources and their oracle transforms, per (log_n, L)
SRC = 4
tables, sources = {}, {}


def shape(log_n, L):
    key = (log_n, L)
    if key not in tables:
        tables[key] = p.U64DcrtTable(log_n, Q61[:L])
        o = oracle.U64DcrtTable(log_n, Q61[:L])
        n = 1 << log_n
        rows = []
        for _ in range(SRC):
            a = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in Q61[:L]])
            f, i = a.copy(), a.copy()
            o.transform_slice(f)
            o.inverse_transform_slice(i)
            rows.append((a, f, i))
        sources[key] = rows
    return tables[key], sources[key]


stats = {"bad": 0, "registered": 0, "zero_copy": 0}
lock = threading.Lock()
# shapes are created up front (table creation is not what is stressed, and the dict is shared by the caller threads)
for log_n in range(6, 17):
    for L in (1, 2, 3):
        shape(log_n, L)


def caller(tid):
    rng = np.random.default_rng(args.seed * 7919 + tid)
    keep = []
    for it in range(args.iters):
        log_n = int(rng.integers(6, 17))
        L = int(rng.integers(1, 4))
        t, rows = shape(log_n, L)
        unit = L << log_n
        batch = int(rng.integers(1, max(1, min(16, args.max_bytes // (unit * 8))) + 1))
        picks = rng.integers(0, SRC, batch)
        inverse = bool(rng.integers(0, 2))
        # a fresh array most of the time (malloc / mmap decides where it lives), sometimes a reused one
        if keep and rng.integers(0, 4) == 0 and keep[-1].size == batch * unit:
            x = keep[-1]
        else:
            x = np.empty(batch * unit, dtype=np.uint64)
        for j, s in enumerate(picks):
            x[j * unit:(j + 1) * unit] = rows[s][0]
        exp = np.concatenate([rows[s][2 if inverse else 1] for s in picks])
        reg = args.register and x.ctypes.data % 16 == 0
        if reg:
            rc = rt.cudaHostRegister(x.ctypes.data, x.nbytes, 0)
            rc = int(getattr(rc, "value", rc))
            if rc != 0:
                print(f"hipHostRegister failed ({rc}) at it={it}", flush=True)
                reg = False
        try:
            (t.inverse_transform_slice if inverse else t.transform_slice)(x)
        finally:
            if reg:
                rt.cudaHostUnregister(x.ctypes.data)
        ok = np.array_equal(x, exp)
        with lock:
            stats["registered"] += reg
            stats["zero_copy"] += reg and x.nbytes <= (1 << 20)
            stats["bad"] += not ok
        if not ok:
            w = np.flatnonzero(x != exp)
            print(f"MISMATCH thread={tid} it={it} logN={log_n} L={L} batch={batch} inverse={inverse} registered={reg} "
                  f"{w.size} bad words of {x.size}, first at {int(w[0])}", flush=True)
        if rng.integers(0, 4) == 0:
            keep.append(x)       # keep some arrays alive so that the heap layout keeps changing
        if len(keep) > 8:
            keep.pop(int(rng.integers(0, len(keep))))


t0 = time.perf_counter()
if args.callers <= 1:
    caller(0)
else:
    cs = [threading.Thread(target=caller, args=(k,)) for k in range(args.callers)]
    [c.start() for c in cs]
    [c.join() for c in cs]
bad, registered_calls, zero_copy_calls = stats["bad"], stats["registered"], stats["zero_copy"]
dt = time.perf_counter() - t0
stop.value = 1
for th in churners:
    th.join()
print(f"{args.iters * max(1, args.callers)} calls in {dt:.1f} s, {bad} mismatches, {registered_calls} caller-registered ({zero_copy_calls} of them "
      f"short enough for the zero-copy kernels), churn threads {args.churn_threads} (damaged blocks {sum(damaged)}), alloc events "
      f"{p.lib().pfhe_debug_alloc_count()}, staging paths "
      f"{[int(p.lib().pfhe_debug_stage_path_count(w)) for w in range(5)]}")
sys.exit(1 if bad or sum(damaged) else 0)
