#!/usr/bin/env python3
"""bench.py — headline benchmark of the primus-fhe hot path on MI355X.

Metric (BASELINE.json): NTT/sec (+ RLWE external-products/sec) at N = 2^16, 3-prime RNS, and the
fraction of the HBM roofline.  One "step" = one forward RNS NTT (U64DcrtTable::transform_slice,
primus_ntt/src/dcrt/prime64.rs:106) over a batch of 4096 RNS polynomials = 12 288 limb-NTTs of
2^16 words, in place, inputs resident in HBM (BASELINE.md config 3').  Independent polynomials
shard across GPUs with no collective: each rank owns its own batch of 4096
("weak" scaling); torch.distributed is used only for the timing barrier.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Every rank also runs the RLWE external product (config 4; batch 1024
per GPU, aggregated over ranks = config 5's scaling curve).  Extra legs on rank 0 at N = 1:
per-kernel HIP-event timing for the roofline object, the fused NTT->mul->INTT rate (config 3), the
u32 tables, and a bounded CPU run of the oracle (the C restatement of the reference's scalar path)
for `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
LOG_N = 16
SEED_NTT, SEED_CONFIG5, SEED_GGSW = 0x5EED000000000003, 0x5EED000000000005, 99
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs (= ranks) of one node; default: WORLD_SIZE when a launcher "
                    "set it, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="RNS polynomials per GPU (default: BASELINE config 3')")
    ap.add_argument("--ext-batch", type=int, default=1024, help="ciphertexts in the external-product leg (config 4)")
    ap.add_argument("--ext-chunk", type=int, default=0, help="ciphertexts per internal pass of the external product")
    ap.add_argument("--ext-total", type=int, default=8192, help="BASELINE config 5: ciphertexts of the FIXED job that is "
                    "split over the ranks (strong scaling); 0 skips that leg")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the timing barrier "
                    "(nccl = RCCL; gloo + --one-device lets two ranks share one GPU for a plumbing check)")
    ap.add_argument("--one-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--skip-device-check", action="store_true", help="when bench.py spawns the ranks itself: do not count "
                    "the visible GPUs first (a rank whose GPU is missing fails by itself and ends its peers)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (test aid: "
                    "exercises the RCCL barrier / max-reduce on a 1-GPU box)")
    ap.add_argument("--dump-dir", default="", help="parity aid for tests/test_gpu_shard.py: every rank writes the HIP "
                    "outputs of its shard (one forward NTT of its RNS polynomials, its config-5 products) as "
                    "rank<r>.npz into this directory; small --batch / --ext-total only")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py spawns the ranks itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus is None:  # `torchrun --nproc-per-node N bench.py` without --gpus
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    return args


def cpu_baseline(seconds: float):
    """All-core throughput of the oracle's forward NTT at N=2^16 (one limb-NTT per task).

    kind = "port": the reference is Rust and cannot be built here.  On an AVX-512 host the
    reference dispatches to its HEXL-style vector kernels (prime64/table.rs:408-418), so the
    baseline runs the oracle's AVX-512 DQ restatement of that path (oracle/pfhe_oracle_avx512.c);
    otherwise, and always for the single-thread scalar figure, the restatement of the scalar path.
    """
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle

    try:
        tmp = tempfile.mkdtemp(prefix="pfhe_oracle_")
        oracle.use_library(oracle.build(native=True, out_dir=tmp))
    except Exception:
        oracle.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container CPU quota (cgroup v2) bounds the usable cores below the visible count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    n = 1 << LOG_N
    tabs = [oracle.U64NttTable(LOG_N, q) for q in Q61]
    rng = np.random.default_rng(1)
    per_task = 32  # ~40 ms of work per task keeps the Python dispatch overhead negligible
    bufs = [rng.integers(0, Q61[i % 3], n * per_task, dtype=np.uint64) for i in range(cores)]

    avx512 = bool(oracle.lib().orc_avx512_available())

    def work(i):  # ctypes releases the GIL
        (tabs[i % 3].transform_slice_avx512 if avx512 else tabs[i % 3].transform_slice)(bufs[i])
        return per_task

    def scalar_work():
        tabs[0].transform_slice(bufs[0])
        return per_task

    # single thread first: scalar path, then the path the threads will run
    t0 = time.perf_counter()
    done1 = 0
    while time.perf_counter() - t0 < min(1.5, seconds / 6):
        done1 += scalar_work()
    single_scalar = done1 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    done1 = 0
    while time.perf_counter() - t0 < min(1.5, seconds / 6):
        done1 += work(0)
    single = done1 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    done = 0
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds * 0.75:
            done += sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    out = {
        "value": done / dt, "unit": "NTT/s", "cores": cores, "kind": "port",
        "single_thread_value": single, "single_thread_scalar_value": single_scalar,
        "backend": "fwd+inv AVX-512 DQ (restatement of prime64/avx512)" if avx512 else "scalar (restatement of prime64/scalar)",
        "sample": f"{done} forward limb-NTTs of N=2^16 (61-bit primes), oracle "
                  f"{'AVX-512 DQ' if avx512 else 'scalar'} Harvey path, {cores} threads, {dt:.1f} s",
    }
    # the reference's own criterion case (benches/bench_u64.rs:8,117-130: q = 1125899906826241 < 2^50, N = 4096), where an
    # AVX-512 IFMA host takes the BIT_SHIFT = 52 rung (prime64/table.rs:166-186): restated in the oracle as well
    try:
        q50, ln50 = 1125899906826241, 12
        t50 = oracle.U64NttTable(ln50, q50)
        ifma = bool(oracle.lib().orc_avx512_ifma_available())
        b50 = [rng.integers(0, q50, (1 << ln50) * 512, dtype=np.uint64) for _ in range(cores)]

        def w50(i, shift):
            t50.transform_batch_avx512(b50[i], shift=shift)
            return 512

        rates = {}
        for label, shift in ((("ifma", 52),) if ifma else ()) + ((("dq", 64),) if avx512 else ()):
            t0 = time.perf_counter()
            d = 0
            with ThreadPoolExecutor(cores) as ex:
                while time.perf_counter() - t0 < 0.5:
                    d += sum(ex.map(lambda i: w50(i, shift), range(cores)))
            rates[label] = d / (time.perf_counter() - t0)
        out["reference_bench_case"] = {
            "workload": "forward NTT, N = 4096, q = 1125899906826241 (primus_ntt/benches/bench_u64.rs:8)", "cores": cores,
            "unit": "NTT/s", "backend_the_reference_dispatches_to": "ifma" if ifma else ("dq" if avx512 else "scalar"), **rates}
    except Exception as e:
        out["reference_bench_case"] = {"error": str(e)[:200]}
    # RLWE external product at the config-4 shape (one ciphertext per task, one shared GGSW): the oracle's
    # restatement of CrtGlwe::mul_dcrt_ggsw_to with its forward transforms on the vector backend when present
    try:
        oracle.lib().orc_set_vector_backend(1 if avx512 else 0)
        dt_tab, base = oracle.U64DcrtTable(LOG_N, Q61), oracle.RNSBase(Q61)
        basis = oracle.BigUintApproxSignedBasis(base, 30)
        glwes = [np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for _ in range(2) for q in Q61])
                 for _ in range(cores)]
        ggsw = np.concatenate([rng.integers(0, q, n, dtype=np.uint64)
                               for _ in range(2 * basis.decompose_length * 2) for q in Q61])

        def ep(i):  # CrtGlwe::mul_dcrt_ggsw_to + DcrtGlwe::into_coeff_form, like the GPU leg
            r = oracle.mul_dcrt_ggsw_to(dt_tab, base, basis, 1, glwes[i], ggsw)
            dt_tab.inverse_transform_slice(r)
            return 1

        ep(0)
        t0 = time.perf_counter()
        done_ep = 0
        with ThreadPoolExecutor(cores) as ex:
            while time.perf_counter() - t0 < max(2.0, seconds * 0.25):
                done_ep += sum(ex.map(ep, range(cores)))
        dte = time.perf_counter() - t0
        out["external_product"] = {
            "value": done_ep / dte, "unit": "RLWE external products/s (coefficient-form output)", "cores": cores,
            "sample": f"{done_ep} products, N=2^16, 3 primes, k=1, ell=6, {cores} threads, {dte:.1f} s"}
        # config 3: NTT -> pointwise product with a shared multiplicand -> INTT, one RNS polynomial per task
        polys = [np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in Q61]) for _ in range(cores)]
        bhat = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in Q61])

        def pm(i):
            dt_tab.transform_slice(polys[i])
            dt_tab.mul_assign(polys[i], bhat)
            dt_tab.inverse_transform_slice(polys[i])
            return 1

        pm(0)
        t0 = time.perf_counter()
        done_pm = 0
        with ThreadPoolExecutor(cores) as ex:
            while time.perf_counter() - t0 < max(1.0, seconds * 0.1):
                done_pm += sum(ex.map(pm, range(cores)))
        dtp = time.perf_counter() - t0
        out["polymul"] = {"value": done_pm / dtp, "unit": "RNS polynomial products/s (NTT+mul+INTT)", "cores": cores,
                          "sample": f"{done_pm} products, N=2^16, 3 primes, {cores} threads, {dtp:.1f} s"}
    except Exception as e:  # the NTT baseline above is the contractual one
        out["external_product"] = {"error": str(e)[:200]}
    finally:
        oracle.lib().orc_set_vector_backend(0)
    return out


def git_blob_hash(path: str) -> str:
    """The id `git hash-object` gives the file: names the exact committed profile a traffic figure was read from."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


_BUILT_KERNELS = None
_BUILT_CODE = None


def built_vgprs(kernel: str):
    """vgpr_count of `kernel` in the libpfhe_hip.so this run loads (its code objects' metadata), or None."""
    global _BUILT_KERNELS
    if _BUILT_KERNELS is None:
        try:
            import primus_fhe_amd as p
            from primus_fhe_amd._codeobj import kernel_resources
            _BUILT_KERNELS = kernel_resources(p.library_path())
        except Exception:
            _BUILT_KERNELS = {}
    r = _BUILT_KERNELS.get(kernel)
    return None if r is None else (r["vgpr"] or 0) + (r["agpr"] or 0)


def built_code_hash(kernel: str):
    """SHA-256 (16 hex digits) of `kernel`'s machine code in the libpfhe_hip.so this run loads, or None."""
    global _BUILT_CODE
    if _BUILT_CODE is None:
        try:
            import primus_fhe_amd as p
            from primus_fhe_amd._codeobj import kernel_code_hashes
            _BUILT_CODE = kernel_code_hashes(p.library_path())
        except Exception:
            _BUILT_CODE = {}
    return _BUILT_CODE.get(kernel)


def provenance(path: str, profiled: dict):
    """A committed counter profile describes the kernels it was taken on.  Its figures are reported only when every
    kernel it covers has, in the library being timed, the MACHINE CODE the profiler ran: the profile names the SHA-256 of
    each kernel's code (tools/pmc_summary.py, tools/collect_profiles2.py through primus-fhe_amd/_codeobj.py), and a
    kernel whose code differs — or a profile that names no hash — makes the profile stale and the traffic null.  (Until
    round 4 the check compared register allocations, which a kernel can keep while it changes what it moves.)
    `profiled`: {kernel: code hash or None}.  Returns (fields for the JSON line, ok)."""
    rows = {}
    ok = bool(profiled)
    for k, h in (profiled or {}).items():
        now = built_code_hash(k)
        rows[k] = {"profiled_code_sha256": h, "built_code_sha256": now, "built_vgpr_count": built_vgprs(k)}
        if h is None or now is None or h != now:
            ok = False
    return {"profile": os.path.basename(path), "profile_git_blob": git_blob_hash(path), "kernels_checked": rows,
            "check": "SHA-256 of each kernel's machine code, profile vs the library being timed",
            "profile_matches_build": ok}, ok


def oracle_pin() -> str:
    """Whether anything executable ties the oracle to the reference's own binaries: tests/golden/reference_digests.json is
    written by integration/emit_golden (cargo) on a host that has a Rust toolchain; until it is
    committed every fixture under tests/golden is this repository's own output and parity stays "unpinned"."""
    path = os.path.join(ROOT, "tests", "golden", "reference_digests.json")
    if not os.path.exists(path):
        return "unpinned (tests/golden/reference_digests.json absent: no Rust toolchain has run integration/emit_golden yet)"
    try:
        d = json.load(open(path))
        return "pinned to the reference's binaries (tests/golden/reference_digests.json: %d cases, reference rev %s)" % (
            len(d.get("cases", [])), d.get("reference_rev", "?"))
    except Exception as e:
        return "unpinned (tests/golden/reference_digests.json unreadable: %s)" % str(e)[:80]


def profile_traffic(name: str, want: dict):
    """HBM bytes per launch of the kernels named `name` whose template arguments match `want` ({position: text}) in the
    newest committed counter profile (profiles/*_rocprof.json: 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes of
    tools/profile_ntt.py), averaged over the launches profiled — reported only when every such kernel has, in the library
    being timed, the machine code the counters were taken on (provenance); None when no profile covers it."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof.json")), reverse=True):
        try:
            rows = json.load(open(path))["kernels"]
        except Exception:
            continue
        sel = []
        for r in rows:
            k = r["kernel"]
            if not k.startswith(name + "<") or r.get("hbm_bytes_per_launch") is None or not r.get("launches"):
                continue
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            if all(i < len(targs) and targs[i] == v for i, v in want.items()):
                sel.append(r)
        if not sel:
            continue
        launches = sum(r["launches"] for r in sel)
        prov, ok = provenance(path, {r["kernel"]: r.get("code_sha256") for r in sel})
        return {"bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel) / launches if ok else None,
                "source": os.path.basename(path), "provenance": prov, "profiled_grids": sorted({r["grid_size"] for r in sel}),
                "method": "2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes (MI355X_MICROARCH.md, HBM), averaged over the "
                          "%d launches profiled" % launches}
    return None


def leg_roofline(kernel: str, launches: int, ms_per_batch: float, alg_bytes_per_batch: float, traffic, copy_gbs=None,
                 note: str | None = None):
    """The roofline object of one leg: the kernel (or form) that runs it, HIP-event / wall time per launch, SURVEY 8d's
    algorithmic bytes per launch, the fraction of the 8 TB/s figure and of the copy rate measured in this run, and the
    HBM bytes per launch from the committed counter profile when its code hash matches the library being timed."""
    avg_ms = ms_per_batch / max(1, launches)
    achieved = alg_bytes_per_batch / (ms_per_batch * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kernel, "launches_per_batch": launches, "avg_launch_ms": avg_ms,
           "algorithmic_bytes_per_launch": alg_bytes_per_batch / max(1, launches), "achieved": achieved,
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "peak_measured": copy_gbs, "frac_of_measured": achieved / copy_gbs if copy_gbs else None,
           "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_unit": "bytes per launch",
           "traffic_source": ("from committed profile " + traffic["source"] + ": " + traffic["method"]) if traffic else None,
           "traffic_provenance": traffic["provenance"] if traffic else None}
    if traffic and traffic["bytes_per_launch"]:
        out["traffic_vs_algorithmic"] = traffic["bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    if note:
        out["note"] = note
    return out


def extprod_traffic(kind: str = "extprod"):
    """Whole-product HBM bytes per external product from the newest committed counter passes
    (profiles/*_extprod_traffic.json — kind "extprod32": *_extprod32_traffic.json, the <u32> product — written by
    tools/collect_profiles2.py) — or None when there is none, or when the kernels it was taken on are not the ones in the
    library being timed (provenance)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % kind)), reverse=True):
        try:
            d = json.load(open(path))
            prov, ok = provenance(path, {k: (d.get("code_sha256_by_kernel") or {}).get(k)
                                         for k in d.get("vgpr_count_by_kernel") or {}})
            return {"bytes_per_product": float(d["bytes_per_product"]) if ok else None, "source": os.path.basename(path),
                    "method": d["method"], "provenance": prov}
        except Exception:
            continue
    return None


def pmc_traffic(kernel: str, batch: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 +
    WRITE_SIZE, separate --pmc runs of tools/profile_ntt.py at this shape; profiles/*_rocprof.json).
    bench.py cannot collect counters itself: the figure is READ FROM A COMMITTED PROFILE, named in `traffic_source` with its
    git blob id, and is null when no profile of this launch shape is committed or when the profiled kernel's machine code
    differs from the one in the library being timed (provenance)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof.json")), reverse=True):
        try:
            rows = json.load(open(path))["kernels"]
        except Exception:
            continue
        want = ("ntt_pipe_" if "pipe" in kernel else "ntt_block_kernel" if "block" in kernel else "ntt_strided_kernel")
        inv = "inv" in kernel
        if want == "ntt_pipe_":  # ntt_pipe_fwd_kernel<A, LOGB> / ntt_pipe_inv_kernel<A, LOGB, MUL>: tiles + 1 launches
            sel = []
            for r in rows:
                k = r["kernel"]
                if (("ntt_pipe_inv_kernel" if inv else "ntt_pipe_fwd_kernel") not in k or "<" not in k
                        or r.get("hbm_bytes_per_launch") is None or not r.get("launches")):
                    continue
                targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
                if targs[0] == "PmArith" and (not inv or targs[2] == "false"):
                    sel.append(r)
            tile_units = -(-batch // 24)
            if sel and max(r["grid_size"] for r in sel) == tile_units * 3 * 16 * 256:
                launches = sum(r["launches"] for r in sel)
                prov, ok = provenance(path, {r["kernel"]: r.get("code_sha256") for r in sel})
                return {"bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel) / launches if ok else None,
                        "source": os.path.basename(path), "provenance": prov,
                        "method": "2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes (MI355X_MICROARCH.md, HBM), "
                                  "averaged over the %d launches profiled" % launches}
            continue
        best = None
        for r in rows:
            k = r["kernel"]
            if want not in k or r.get("hbm_bytes_per_launch") is None or "<" not in k:
                continue
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            if targs[0] == "B32Arith":
                continue  # the u32 tables' instantiation of the same kernel
            if "block" in want:  # <A, LOGB, INV, MUL>
                is_inv = len(targs) > 2 and targs[2] == "true"
                if len(targs) > 3 and targs[3] == "true":
                    continue  # fused-product variant reads a second operand
            else:                # <A, K, VEC, INV, FINAL>
                is_inv = len(targs) > 3 and targs[3] == "true"
            if is_inv != inv:
                continue
            if best is None or r["grid_size"] > best["grid_size"]:
                best = r
        # only a launch of the same shape counts: block pass = 16 workgroups of 256 per polynomial,
        # strided pass = N/32 threads per polynomial
        want_grid = batch * 3 * ((1 << LOG_N) // 16 if "block" in want else (1 << LOG_N) // 32)
        if best and best["grid_size"] == want_grid:
            prov, ok = provenance(path, {best["kernel"]: best.get("code_sha256")})
            return {"bytes_per_launch": best["hbm_bytes_per_launch"] if ok else None, "source": os.path.basename(path),
                    "provenance": prov,
                    "method": "2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes (MI355X_MICROARCH.md, HBM)"}
    return None


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes (one per GPU, the
    environment torch.distributed.run would give them) and relay rank 0's JSON line.  The parent never execs and runs
    no GPU work; counting the visible devices (skipped under --skip-device-check or --one-device) may initialise the HIP
    runtime in the parent, which is harmless because every rank is a fresh child process."""
    import socket
    import subprocess

    if not args.one_device and not args.skip_device_check:
        import torch
        have = torch.cuda.device_count()  # (falls back to hipGetDeviceCount when amdsmi is unavailable)
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --one-device --dist-backend gloo for a "
                             "plumbing check on one GPU)" % (args.gpus, have))
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting at a barrier: end them (by PID, never by pattern) instead of hanging
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.terminate()
                    try:
                        rcs[i] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[i] = pr.wait()
            break
        time.sleep(0.2)
    return max((abs(rc) for rc in rcs if rc is not None), default=0)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    import primus_fhe_amd as p
    from primus_fhe_amd._lib import check, u64p

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); use gpurun")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    from primus_fhe_amd.shard import device_identity, dist_evidence
    # what the process group saw (not what the launcher's environment says): filled into the line below
    dist_seen = dist_evidence(dist, device_identity(local_rank)) if dist is not None else None

    n, L, batch = 1 << LOG_N, 3, args.batch
    words = batch * L * n
    dump = {}
    table = p.U64DcrtTable(LOG_N, Q61, device=local_rank)
    ep_batch = min(args.ext_batch, batch)
    xbuf = torch.empty(max(words, ep_batch * 2 * L * n), dtype=torch.int64, device="cuda")  # also the config-4 input
    x = xbuf[:words]
    mods = np.array(Q61, np.uint64)
    from primus_fhe_amd.shard import fill_job_shard, timed_steps

    # "also report against a measured device-copy bandwidth" — one device-to-device copy of half the resident
    # buffer into the other half (3 GiB read + 3 GiB written at the default batch), by the runtime's copy, by torch's
    # element-wise copy kernel and by the library's 16-bytes-per-lane non-temporal copy kernel, HIP events on the launch
    # stream; the best one is `peak_measured`
    copy_bw = None
    if rank == 0 and not args.dump_dir and words >= (1 << 24):
        half = (words // 2) & ~1
        src, dst = x[:half], x[half:2 * half]
        st = torch.cuda.current_stream()

        def timed_copy(fn, reps=6):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            e1.synchronize()
            return 2 * half * 8 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9

        rt = timed_copy(lambda: check(p.lib().pfhe_memcpy_d2d(local_rank, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                                               half * 8, C.c_void_p(st.cuda_stream))))
        tk = timed_copy(lambda: dst.copy_(src))
        # this library's own streaming shape: one 16-byte vector per thread, non-temporal loads and stores
        sk = timed_copy(lambda: check(p.lib().pfhe_stream_copy_dev(local_rank, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                                                    half * 8, C.c_void_p(st.cuda_stream))))
        copy_bw = {"GBps": max(rt, tk, sk), "hipMemcpyDtoDAsync_GBps": rt, "torch_copy_kernel_GBps": tk,
                   "stream_copy_kernel_16B_per_lane_nontemporal_GBps": sk,
                   "bytes_moved_per_copy": 2 * half * 8, "method": "read + written bytes / HIP-event time, 6 copies back to back"}
    copy_gbs = copy_bw["GBps"] if copy_bw else None

    # rank r owns RNS polynomials [r*batch, (r+1)*batch) of a job of world*batch: its input depends on the position
    # in the job only, so the union over ranks is the same job whatever the world size
    fill_job_shard(p.lib(), local_rank, x.data_ptr(), rank * batch, batch, L * n, Q61, n, SEED_NTT)

    def step():
        # forward transform of a canonical batch; the output of one step (canonical, bit-reversed
        # order) is a valid input of the next, so the timed loop needs no re-initialisation
        table.transform_dev(x)

    dt = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dist,
                     "cuda" if args.dist_backend == "nccl" else "cpu")

    limb_ntts = world * batch * L * args.steps
    value = limb_ntts / dt
    result = {
        "metric": "NTT/sec at N=2^16, 3-prime RNS (forward limb-NTTs, batch %d RNS polynomials per GPU)" % batch,
        "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2] shape: N=2^16, 3-prime RNS (61-bit), batch=%d per GPU, "
                               "forward DCRT NTT in place (config 3')" % batch,
                   "log_n": LOG_N, "moduli": Q61, "batch_per_gpu": batch, "sharding": "batch, no collectives",
                   # the headline rate is specific to this shape of prime: q = 2^K - c with c < 2^(K-33) takes the
                   # pseudo-Mersenne butterflies (csrc/pfhe_ntt_device.hpp PmArith); any other q < 2^61 the Montgomery-form
                   # ones — `roofline_generic` below is that rate on the same data
                   "prime_shape": "pseudo-Mersenne (q = 2^61 - c, c < 2^28: every prime of SURVEY 8c's 61-bit triple)"},
        "oracle_pin": oracle_pin(),
        "device_copy": copy_bw,
        "hbm_roofline_frac": value * 16 * n / world / (HBM_PEAK_GBS * 1e9),
        # torch.distributed's own view: backend, world size, and every rank's device (PCI address, UUID) gathered over the
        # group; without a process group (one rank) the one device this process used
        "dist": dist_seen if dist_seen is not None else {"backend": None, "world_size": 1, "devices": [dict(device_identity(local_rank), rank=0)],
                                                         "distinct_devices": 1},
    }
    if dist_seen is not None and dist_seen["world_size"] != world:
        raise SystemExit("bench.py: the process group has %d ranks, the launcher said %d" % (dist_seen["world_size"], world))

    # ---- config 4 / 5: RNS gadget external product, k=1, logB=30 (ell=6), batch 1024 per GPU, one shared
    #      GGSW replicated per device; every rank runs it, the rate is aggregated over ranks (weak scaling,
    #      no collective on the data path) ----
    base = p.RNSBase(Q61, device=local_rank)
    basis = p.BigUintApproxSignedBasis(base, 30)
    ctx = p.DcrtGlevContext(table, base, basis, 1, args.ext_chunk)
    glwe_words, ggsw_words = ep_batch * 2 * L * n, ctx.ggsw_len()
    check(p.lib().pfhe_fill_uniform_dev(local_rank, C.c_void_p(xbuf.data_ptr()), glwe_words, mods.ctypes.data_as(u64p), L, n,
                                        0x5EED000000000004 + rank, None))
    glwe = xbuf[:glwe_words]                           # canonical residues: a valid CrtGlwe batch
    ggsw = torch.empty(ggsw_words, dtype=torch.int64, device="cuda")
    check(p.lib().pfhe_fill_uniform_dev(local_rank, C.c_void_p(ggsw.data_ptr()), ggsw_words, mods.ctypes.data_as(u64p), L,
                                        n, SEED_GGSW, None))
    out = torch.empty(glwe_words, dtype=torch.int64, device="cuda")
    ep_steps = max(2, args.steps // 3)
    dte = timed_steps(lambda: p.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, into_coeff_form=True), ep_steps, 1,
                      torch.cuda.synchronize, dist, "cuda" if args.dist_backend == "nccl" else "cpu") / ep_steps
    result["external_product"] = {
        "value": world * ep_batch / dte, "unit": "RLWE external products/s (CrtGlwe x DcrtGgsw -> coefficient form), "
                                                 "whole job", "n_gpus": world,
        "batch_per_gpu": ep_batch, "ms_per_batch": dte * 1e3, "gadget": {"log_basis": 30, "ell": 6, "k": 1},
        "ggsw": "one shared 36 MiB DcrtGgsw per GPU", "chunk": args.ext_chunk or "default (128 at this shape)",
        "hbm_roofline_frac": ep_batch / dte * 96 * n / (HBM_PEAK_GBS * 1e9),
        "limb_ntts_per_product": 42}
    # kernel groups of the product, HIP events on the launch stream (rank 0): which one dominates, and its share of
    # the HBM roofline on its algorithmic bytes (digits of (k+1)*ell*L polynomials in, (k+1)*L polynomials out)
    if rank == 0:
        try:
            s_cur = torch.cuda.current_stream()
            ms_dec, ms_mac, launches = p.profile_mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx, stream=s_cur.cuda_stream)
            dom_ms = max(ms_dec, ms_mac)
            mac_bytes = ep_batch * (2 * 6 * L + 2 * L) * n * 8  # transformed digits read + result written (key: L2)
            dec_bytes = ep_batch * (2 * L + 2 * 6 * L) * n * 8  # CRT polynomials read + strided-pass digits written
            dom_bytes = mac_bytes if ms_mac >= ms_dec else dec_bytes
            traffic = extprod_traffic()
            alg_bytes_ep = 96 * n * ep_batch  # CrtGlwe in (48 N) + DcrtGlwe out (48 N) per product
            result["external_product"]["roofline"] = {
                "bound": "hbm", "kernel": "gadget_block_mulacc_kernel (block pass of the digits' transform + multiply-accumulate"
                                          " + inverse block pass of the result)"
                if ms_mac >= ms_dec else "gadget_signed_digits_kernel + digits_strided_kernel",
                "avg_launch_ms": dom_ms / max(1, launches), "launches_per_batch": launches,
                "ms_per_batch": {"digits_and_strided_pass": ms_dec, "block_pass_and_multiply_accumulate": ms_mac},
                # the PRODUCT against the roofline, on SURVEY §8d's algorithmic bytes and the whole batch's time
                "algorithmic_bytes_per_batch": alg_bytes_ep,
                "achieved": alg_bytes_ep / dte / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes_ep / dte / 1e9 / HBM_PEAK_GBS,
                "peak_measured": copy_gbs, "frac_of_measured": alg_bytes_ep / dte / 1e9 / copy_gbs if copy_gbs else None,
                # the dominant KERNEL on the bytes the two-pass plan makes it move (transformed digits in + result out): a
                # statement about that kernel under this plan, not about the product
                "kernel_compulsory_bytes": dom_bytes,
                "kernel_achieved": dom_bytes / (dom_ms * 1e-3) / 1e9,
                "kernel_frac": dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # whole product, every kernel: HBM bytes per product from the committed counter passes x this batch
                "traffic": traffic["bytes_per_product"] * ep_batch if traffic and traffic["bytes_per_product"] else None,
                "traffic_unit": "bytes per batch of %d products, all kernels of the product (coefficient form)" % ep_batch,
                "traffic_source": ("from committed profile " + traffic["source"] + ": " + traffic["method"]) if traffic else None,
                "traffic_provenance": traffic["provenance"] if traffic else None,
                "traffic_vs_algorithmic": traffic["bytes_per_product"] / (96 * n) if traffic and traffic["bytes_per_product"] else None,
                "note": "the kernel runs 12 forward block transforms + 72 multiply-accumulates per word + 2 inverse block "
                        "transforms per output block: VALU-bound (VALUBusy 90 %), and the plan moves ~9x the product's "
                        "algorithmic bytes (the 36 half-transformed digit polynomials are written and read once)"}
        except Exception as e:  # measurement aid only
            result["external_product"]["roofline"] = {"error": str(e)[:200]}
    del out
    # ---- config 5 as BASELINE.md defines it: a FIXED job of --ext-total ciphertexts split over the ranks ----
    if args.ext_total > 0:
        from primus_fhe_amd.shard import shard_range, strong_scaling_leg
        b5, e5 = shard_range(args.ext_total, world, rank)
        mine = e5 - b5
        g5 = torch.empty(max(1, mine) * 2 * L * n, dtype=torch.int64, device="cuda")
        fill_job_shard(p.lib(), local_rank, g5.data_ptr(), b5, mine, 2 * L * n, Q61, n, SEED_CONFIG5)
        o5 = torch.empty_like(g5)

        def run_shard(begin, end):
            if end > begin:
                p.mul_dcrt_ggsw_to_dev(g5[:(end - begin) * 2 * L * n], ggsw, o5[:(end - begin) * 2 * L * n], ctx,
                                       into_coeff_form=True)

        leg = strong_scaling_leg(args.ext_total, world, rank, run_shard, max(2, args.steps // 5), 1,
                                 torch.cuda.synchronize, dist, "cuda" if args.dist_backend == "nccl" else "cpu")
        leg.update({"unit": "RLWE external products/s, whole job (BASELINE config 5: batch split over the GPUs, "
                            "no collective on the data path)",
                    "hbm_roofline_frac": leg["value"] / world * 96 * n / (HBM_PEAK_GBS * 1e9)})
        result["external_product_config5"] = leg
        if args.dump_dir:
            dump["config5_range"] = np.array([b5, e5])
            dump["config5_out"] = o5[:mine * 2 * L * n].cpu().numpy().view(np.uint64)
        del g5, o5
    del ggsw, ctx
    if args.dump_dir:
        if batch > 64 or args.ext_total > 64:
            raise SystemExit("--dump-dir is for small parity runs (--batch, --ext-total <= 64)")
        fill_job_shard(p.lib(), local_rank, x.data_ptr(), rank * batch, batch, L * n, Q61, n, SEED_NTT)
        table.transform_dev(x)
        dump["ntt_range"] = np.array([rank * batch, (rank + 1) * batch])
        dump["ntt_out"] = x.cpu().numpy().view(np.uint64)
        os.makedirs(args.dump_dir, exist_ok=True)
        np.savez(os.path.join(args.dump_dir, "rank%d.npz" % rank), **dump)

    if rank == 0 and not args.dump_dir:
        # ---- per-kernel timing (HIP events on the launch stream) for the roofline object: rank 0's GPU, any N ----
        npass = p.lib().pfhe_dcrt_transform_num_passes(table._h)
        stream = torch.cuda.current_stream()
        per_pass = []
        for i in range(npass):
            name = p.lib().pfhe_dcrt_transform_pass_name(table._h, 0, i).decode()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, args.steps)
            check(p.lib().pfhe_dcrt_transform_pass_dev(table._h, C.c_void_p(x.data_ptr()), words, 0, i, 0,
                                                       C.c_void_p(stream.cuda_stream)))
            e0.record(stream)
            for _ in range(reps):
                check(p.lib().pfhe_dcrt_transform_pass_dev(table._h, C.c_void_p(x.data_ptr()), words, 0, i, 0,
                                                           C.c_void_p(stream.cuda_stream)))
            e1.record(stream)
            e1.synchronize()
            per_pass.append((name, e0.elapsed_time(e1) / reps))
        dom = max(per_pass, key=lambda t: t[1])
        alg_bytes = 16 * n * batch * L  # each pass reads and writes every coefficient once
        achieved = alg_bytes / (dom[1] * 1e-3) / 1e9
        pmc = pmc_traffic(dom[0], batch)
        standalone = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      # HBM bytes per launch of this kernel from the committed PMC passes (null if none match)
                      "traffic": pmc["bytes_per_launch"] if pmc else None,
                      "traffic_unit": "bytes per launch",
                      "traffic_source": ("from committed profile " + pmc["source"] + ": " + pmc["method"]) if pmc else None,
                      "traffic_provenance": pmc["provenance"] if pmc else None,
                      "avg_launch_ms": dom[1], "algorithmic_bytes_per_launch": alg_bytes,
                      "peak_measured": copy_gbs, "frac_of_measured": achieved / copy_gbs if copy_gbs else None,
                      "note": "fraction of the HBM roofline as the metric asks; the kernel's own limit is the "
                              "integer ALU and its LDS / twiddle traffic (profiles/r02_*), not HBM"}
        result["kernels_ms"] = {k: v for k, v in per_pass}
        form, launches = table.transform_form(words)
        if form.startswith("ntt_pipe_"):
            # The timed step is tiles + 1 back-to-back launches of ONE kernel (block pass of tile k-1 and strided pass of
            # tile k in each workgroup): time them as the step does, HIP events on the launch stream.  One launch's
            # algorithmic bytes = the transform's 16*N bytes per limb-polynomial over the launches (each coefficient is
            # read once and written once by the TRANSFORM; the two-pass plan moves it twice, which `traffic` shows).
            check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 5, None))
            table.transform_dev(x)
            reps = max(3, args.steps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                table.transform_dev(x)
            e1.record(stream)
            e1.synchronize()
            ms_launch = e0.elapsed_time(e1) / reps / launches
            achieved_p = alg_bytes / launches / (ms_launch * 1e-3) / 1e9
            pmc_p = pmc_traffic(form, batch)
            result["roofline"] = {"bound": "hbm", "kernel": form, "achieved": achieved_p, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": achieved_p / HBM_PEAK_GBS,
                                  "traffic": pmc_p["bytes_per_launch"] if pmc_p else None,
                                  "traffic_unit": "bytes per launch",
                                  "traffic_source": ("from committed profile " + pmc_p["source"] + ": " + pmc_p["method"]) if pmc_p else None,
                                  "traffic_provenance": pmc_p["provenance"] if pmc_p else None,
                                  "avg_launch_ms": ms_launch, "launches_per_step": launches,
                                  "algorithmic_bytes_per_launch": alg_bytes / launches,
                                  "peak_measured": copy_gbs, "frac_of_measured": achieved_p / copy_gbs if copy_gbs else None,
                                  "note": "the step's only kernel; it moves every coefficient twice (two-pass plan), so "
                                          "its HBM floor is 2x the algorithmic bytes; it runs AT the 1400 W package power cap "
                                          "with the shader clock throttled to ~1.9 GHz (profiles/r02_power_probe.txt); "
                                          "stand-alone passes in roofline_passes"}
            result["roofline_passes"] = standalone
        else:
            result["roofline"] = standalone
    # the remaining legs reuse the resident batch buffer at the BASELINE shapes (config 2 alone needs 512 MiB of it):
    # they run at the default batch only; reduced --batch runs (tests) stop at the roofline object
    if rank == 0 and world == 1 and not args.dump_dir and batch >= 4096:
        # ---- the same step sustained for about three seconds: the timed K steps above last ~0.1 s, the chip sits at its
        #      power cap under these kernels and settles its clocks over seconds; this is the steady-state rate (and a
        #      GPU-busy window long enough for an outside sampler to see) ----
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 5, None))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sus = 0
        while time.perf_counter() - t0 < 3.0:
            for _ in range(50):
                table.transform_dev(x)
            torch.cuda.synchronize()
            n_sus += 50
        dt_sus = time.perf_counter() - t0
        result["sustained"] = {"seconds": dt_sus, "steps": n_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                               "value": batch * L * n_sus / dt_sus, "unit": "NTT/s (the headline step, back to back)",
                               "hbm_roofline_frac": batch * L * n_sus / dt_sus * 16 * n / (HBM_PEAK_GBS * 1e9)}
        # the single-pass loops above left x in an arbitrary state: restore canonical residues
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 5, None))
        # ---- the inverse transform at the same shape (U64DcrtTable::inverse_transform_slice, prime64/table.rs:560) ----
        def time_ms(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        reps = max(3, args.steps)
        ms_inv = time_ms(lambda: table.inverse_transform_dev(x), reps)
        form_i, launches_i = table.transform_form(words, inverse=True)
        result["intt"] = {"value": batch * L / ms_inv * 1e3, "unit": "NTT/s (inverse limb-NTTs, N=2^16, 3 primes)",
                          "ms_per_batch": ms_inv, "hbm_roofline_frac": batch * L / ms_inv * 1e3 * 16 * n / (HBM_PEAK_GBS * 1e9),
                          "roofline": leg_roofline(form_i + "<PmArith, 12, false>", launches_i, ms_inv, 16 * n * batch * L,
                                                   profile_traffic("ntt_pipe_inv_kernel", {0: "PmArith", 2: "false"})
                                                   if form_i.startswith("ntt_pipe_") else None, copy_gbs)}
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 6, None))
        # ---- the generic-prime path on the same data: what any q < 2^61 that is not of pseudo-Mersenne shape gets
        #      (MontArith: one-word Montgomery product, 7 multiplies); and the reference's own Shoup scheme (10 multiplies),
        #      which primes in [2^61, 2^62) still take ----
        os.environ["PFHE_DISABLE_PM"] = "1"  # read once, when the table is created
        try:
            table_shoup = p.U64DcrtTable(LOG_N, Q61, device=local_rank)
            os.environ["PFHE_DISABLE_MONT"] = "1"
            table_shoup10 = p.U64DcrtTable(LOG_N, Q61, device=local_rank)
        finally:
            del os.environ["PFHE_DISABLE_PM"]
            os.environ.pop("PFHE_DISABLE_MONT", None)
        ms_sh = time_ms(lambda: table_shoup.transform_dev(x), reps)
        ms_shi = time_ms(lambda: table_shoup.inverse_transform_dev(x), reps)
        ms_s10 = time_ms(lambda: table_shoup10.transform_dev(x), reps)
        result["ntt_generic_prime"] = {
            "value": batch * L / ms_sh * 1e3, "unit": "NTT/s (forward limb-NTTs, MontArith: any q < 2^61)",
            "ms_per_batch": ms_sh, "inverse_ms_per_batch": ms_shi, "shoup_form_ms_per_batch": ms_s10,
            "hbm_roofline_frac": batch * L / ms_sh * 1e3 * 16 * n / (HBM_PEAK_GBS * 1e9)}
        form_g, launches_g = table_shoup.transform_form(words)
        result["roofline_generic"] = leg_roofline(
            form_g + "<MontArith, 12>", launches_g, ms_sh, 16 * n * batch * L,
            profile_traffic("ntt_pipe_fwd_kernel", {0: "MontArith"}) if form_g.startswith("ntt_pipe_") else None, copy_gbs,
            "the headline step on the same data through the arithmetic any q < 2^61 that is NOT of pseudo-Mersenne shape "
            "gets (Montgomery-form butterflies, 7 multiplies against 6): the shape dependence of `roofline.frac`")
        del table_shoup10
        del table_shoup
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 5, None))
        # ---- BASELINE config 2: N = 2^14, one 61-bit prime, batch 4096 (single block pass: one HBM read + write) ----
        n14, b14 = 1 << 14, 4096
        t14 = p.U64NttTable(14, Q61[0], device=local_rank)
        x14 = x[:b14 * n14]
        m14 = np.array(Q61[:1], np.uint64)
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x14.data_ptr()), b14 * n14, m14.ctypes.data_as(u64p), 1, n14, 14, None))
        ms14 = time_ms(lambda: t14.transform_dev(x14), 4 * reps)
        ms14i = time_ms(lambda: t14.inverse_transform_dev(x14), 4 * reps)
        d14 = p.U64DcrtTable(14, Q61[:1], device=local_rank)  # the same kernels behind the DCRT handle: names the form
        f14, l14 = d14.transform_form(b14 * n14)
        f14i, l14i = d14.transform_form(b14 * n14, inverse=True)
        pk = lambda inv: profile_traffic("ntt_persist_kernel", {0: "PmArith", 1: "14", 2: "true" if inv else "false"})
        result["ntt_2p14"] = {
            "workload": "BASELINE configs[1]: N=2^14, q=%d, batch=%d, in place" % (Q61[0], b14),
            "forward": {"value": b14 / ms14 * 1e3, "unit": "NTT/s", "ms_per_batch": ms14,
                        "hbm_roofline_frac": b14 / ms14 * 1e3 * 16 * n14 / (HBM_PEAK_GBS * 1e9),
                        "roofline": leg_roofline(f14, l14, ms14, 16 * n14 * b14, pk(False) if "persist" in f14 else None, copy_gbs,
                                                 "single pass: every coefficient read once and written once")},
            "inverse": {"value": b14 / ms14i * 1e3, "unit": "NTT/s", "ms_per_batch": ms14i,
                        "hbm_roofline_frac": b14 / ms14i * 1e3 * 16 * n14 / (HBM_PEAK_GBS * 1e9),
                        "roofline": leg_roofline(f14i, l14i, ms14i, 16 * n14 * b14, pk(True) if "persist" in f14i else None, copy_gbs)}}
        del t14, d14
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 5, None))
        # ---- config 3: fused NTT -> pointwise mul (shared multiplicand) -> INTT ----
        bhat = torch.empty(L * n, dtype=torch.int64, device="cuda")
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(bhat.data_ptr()), L * n, mods.ctypes.data_as(u64p), L, n,
                                            77, None))
        table.mul_dcrt_polynomial_dev(x, bhat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(2, args.steps // 2)
        for _ in range(reps):
            table.mul_dcrt_polynomial_dev(x, bhat)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / reps
        result["polymul"] = {"value": batch / dtp, "unit": "RNS polynomial products/s (NTT+mul+INTT, shared multiplicand)",
                             "ms_per_batch": dtp * 1e3, "hbm_roofline_frac": batch / dtp * 48 * n / (HBM_PEAK_GBS * 1e9),
                             "algorithmic_bytes_per_product": 48 * n,
                             "roofline": leg_roofline("ntt_pipe_fwd_kernel + ntt_pipe_mid_kernel + ntt_pipe_inv_kernel (strided | "
                                                      "block-product-block | strided)", 1, dtp * 1e3, 48 * n * batch, None, copy_gbs)}
        # BASELINE.md's primary form of config 3: a PER-ELEMENT multiplicand (rlwe/crt.rs:42-65) — read a 24N + read b-hat
        # 24N + write 24N = 72 N bytes per product; a second resident operand of the batch's size
        bfull = torch.empty(words, dtype=torch.int64, device="cuda")
        check(p.lib().pfhe_fill_uniform_dev(0, C.c_void_p(bfull.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, 78, None))
        table.mul_dcrt_polynomial_dev(x, bfull)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            table.mul_dcrt_polynomial_dev(x, bfull)
        torch.cuda.synchronize()
        dtq = (time.perf_counter() - t0) / reps
        result["polymul_per_element"] = {
            "value": batch / dtq, "unit": "RNS polynomial products/s (NTT+mul+INTT, one multiplicand per element)",
            "ms_per_batch": dtq * 1e3, "hbm_roofline_frac": batch / dtq * 72 * n / (HBM_PEAK_GBS * 1e9),
            "algorithmic_bytes_per_product": 72 * n,
            "roofline": leg_roofline("ntt_pipe_fwd_kernel + ntt_pipe_mid_kernel + ntt_pipe_inv_kernel", 1, dtq * 1e3,
                                     72 * n * batch, None, copy_gbs)}
        del bfull
        # ---- §8f rank 1: the u32 / low-q tables at the same shape (N=2^16, three 30-bit primes) ----
        q30 = [1073479681, 1071513601, 1070727169]
        t32 = p.U32DcrtTable(LOG_N, q30, device=local_rank)
        x32 = x.view(torch.int32)[:words]           # reuse the first half of the resident buffer
        t32.fill_uniform_dev(x32, 0x5EED000000000032)
        t32.transform_dev(x32)
        torch.cuda.synchronize()
        reps = max(3, args.steps)
        t0 = time.perf_counter()
        for _ in range(reps):
            t32.transform_dev(x32)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / reps
        ms32i = time_ms(lambda: t32.inverse_transform_dev(x32), reps)   # (pipelined form from 1 GiB of data, round 5)
        f32, l32 = t32.transform_form(words)          # pipelined form from 1 GiB of data: tiles of 512 MiB + 1 launches
        f32i, l32i = t32.transform_form(words, inverse=True)
        result["ntt_u32"] = {"value": batch * L / dt32, "unit": "NTT/s (forward limb-NTTs, u32 data, 30-bit primes)",
                             "ms_per_batch": dt32 * 1e3, "moduli": q30,
                             "hbm_roofline_frac": batch * L / dt32 * 8 * n / (HBM_PEAK_GBS * 1e9),
                             "roofline": leg_roofline(f32 + "<B32Arith, 11>", l32, dt32 * 1e3, 8 * n * batch * L,
                                                      profile_traffic("ntt_pipe_fwd_kernel", {0: "B32Arith"}), copy_gbs),
                             "inverse": {"value": batch * L / ms32i * 1e3, "unit": "NTT/s", "ms_per_batch": ms32i,
                                         "hbm_roofline_frac": batch * L / ms32i * 1e3 * 8 * n / (HBM_PEAK_GBS * 1e9),
                                         "roofline": leg_roofline(f32i + "<B32Arith, 11, false>", l32i, ms32i, 8 * n * batch * L,
                                                                  profile_traffic("ntt_pipe_inv_kernel", {0: "B32Arith", 2: "false"}),
                                                                  copy_gbs)}}
        # ---- the <u32> external product (CrtGlwe<u32> x DcrtGgsw over U32DcrtTable): three 30-bit primes, log B = 15 ->
        #      ell = 6, k = 1, batch 1024, one shared GGSW; algorithmic bytes = CrtGlwe in (24 N) + DcrtGlwe out (24 N) ----
        base32 = p.RNSBase32(q30, device=local_rank)
        basis32 = p.BigUintApproxSignedBasis32(base32, 15)
        ctx32 = p.DcrtGlevContext32(t32, base32, basis32, 1)
        eb32 = min(args.ext_batch, batch)
        g32 = x32[:eb32 * 2 * L * n]
        t32.fill_uniform_dev(g32, 0x5EED000000000432)
        k32 = torch.empty(ctx32.ggsw_len(), dtype=torch.int32, device="cuda")
        t32.fill_uniform_dev(k32, 99)
        o32 = torch.empty_like(g32)
        ms_ep32 = time_ms(lambda: p.mul_dcrt_ggsw_to_dev(g32, k32, o32, ctx32, into_coeff_form=True), max(2, args.steps // 3))
        alg32 = 48 * n * eb32
        result["external_product_u32"] = {
            "value": eb32 / ms_ep32 * 1e3, "unit": "RLWE external products/s (CrtGlwe<u32> x DcrtGgsw over U32DcrtTable -> "
                                                     "coefficient form)",
            "batch": eb32, "ms_per_batch": ms_ep32, "moduli": q30, "gadget": {"log_basis": 15, "ell": basis32.decompose_length(), "k": 1},
            "algorithmic_bytes_per_product": 48 * n, "hbm_roofline_frac": alg32 / (ms_ep32 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline": leg_roofline("gadget_signed_digits_kernel<u32> + digits_strided32_kernel + gadget_block_mulacc32_kernel "
                                     "(block pass + multiply-accumulate + inverse block pass) + inverse strided pass", 1, ms_ep32,
                                     alg32, None, copy_gbs,
                                     "the 64-bit plan's kernels on B32Arith words (two u32 coefficients per 64-bit word): the "
                                     "36 half-transformed digit polynomials are written and read once, their transforms stay "
                                     "on chip")}
        tr32 = extprod_traffic("extprod32")   # whole product, every kernel, from the committed counter passes
        if tr32:
            rf = result["external_product_u32"]["roofline"]
            rf["traffic"] = tr32["bytes_per_product"] * eb32 if tr32["bytes_per_product"] else None
            rf["traffic_unit"] = "bytes per batch of %d products, all kernels of the product (coefficient form)" % eb32
            rf["traffic_source"] = "from committed profile " + tr32["source"] + ": " + tr32["method"]
            rf["traffic_provenance"] = tr32["provenance"]
            rf["traffic_vs_algorithmic"] = tr32["bytes_per_product"] / (48 * n) if tr32["bytes_per_product"] else None
        del k32, o32, ctx32
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
