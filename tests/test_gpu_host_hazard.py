"""The host-memory hazard of round 4 (profiles/r04_experiments.txt item 6), driven through the PUBLIC ABI.

Round 4's library registered the caller's pageable slice for the duration of a call (hipHostRegister -> kernels on the
mapped range -> hipHostUnregister) and, about once in twenty full test runs, produced wrong words and — under a debugging
allocator — a corrupted heap; the registration was withdrawn, cause not established.  The library still uses memory the
CALLER pinned as it is, so a caller who registers per call recreates that configuration.  tools/stress_host_slice.py does
exactly that: >= 20 000 calls, every slice registered by the caller for its call, fresh arrays that reappear at reused
addresses, host threads (tests/native/heap_churn.c) growing and trimming malloc arenas and mapping / unmapping large blocks
meanwhile, every result compared with the oracle's transform and every churned block checked for damage.  The reference's
contract is `&self, &mut [T]` on any memory (primus_ntt/src/ntt/prime64/table.rs:541-563)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "stress_host_slice.py")


def run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, TOOL] + args, capture_output=True, text=True, timeout=timeout, env=env)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    line = [l for l in r.stdout.splitlines() if " calls in " in l][-1]
    assert " 0 mismatches" in line and "damaged blocks 0" in line, line
    return line


def test_caller_registered_slices_under_heap_churn_20000_calls():
    line = run(["20000", "--register", "--churn-threads", "3"])
    assert "20000 caller-registered" in line
    zero_copy = int(line.split("caller-registered (")[1].split()[0])
    assert zero_copy >= 10000, line       # most of them on the zero-copy kernels: the configuration that failed


def test_caller_registered_slices_under_a_debugging_allocator():
    """MALLOC_CHECK_=3 keeps large arrays on the heap (addresses are reused from call to call) and aborts at the first
    damaged heap header."""
    run(["8000", "--register", "--churn-threads", "3", "--seed", "7"], {"MALLOC_CHECK_": "3"})


def test_pageable_slices_under_heap_churn_including_the_helper_thread_form():
    """The library's own paths (pool buffer, pageable copies, and — slices of 8 MiB and more — the helper-thread form)
    under the same churn, four caller threads through the same handles."""
    run(["6000", "--churn-threads", "3", "--seed", "9"])
    run(["600", "--churn-threads", "2", "--seed", "11", "--max-bytes", str(24 << 20), "--callers", "4"])
    # the multi-threaded setting under the debugging allocator, the one that triggered round 4's item 6
    run(["300", "--churn-threads", "2", "--seed", "13", "--max-bytes", str(24 << 20), "--callers", "4"], {"MALLOC_CHECK_": "3"})


def test_library_side_registration_is_opt_in_and_exact_when_asked_for():
    """PFHE_STAGE_REGISTER_PAGEABLE=1 (off by default — DESIGN.md §5 says why) makes the library register a pageable slice
    for its call, as round 4's did: path counter 5 moves, every word still matches; without the variable it never moves."""
    on = run(["1500", "--churn-threads", "2", "--seed", "17"], {"PFHE_STAGE_REGISTER_PAGEABLE": "1"})
    assert eval(on.split("staging paths ")[1])[5] > 1000, on
    off = run(["300", "--churn-threads", "2", "--seed", "17"], {"PFHE_STAGE_REGISTER_PAGEABLE": "0"})
    assert eval(off.split("staging paths ")[1])[5] == 0, off
