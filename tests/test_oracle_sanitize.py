"""AddressSanitizer + UBSan over the oracle's C code (CPU build only; GPU sanitizers are not
available on the pool): oracle/sanitize_main.c drives every entry point at small sizes."""
import os
import shutil
import subprocess
import tempfile

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_under_asan_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sanitize")
        cmd = ["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
               "-fno-omit-frame-pointer", "-I", ORACLE, os.path.join(ORACLE, "sanitize_main.c"),
               os.path.join(ORACLE, "pfhe_oracle.c"), os.path.join(ORACLE, "pfhe_oracle_avx512.c"),
               os.path.join(ORACLE, "pfhe_oracle_rns32.c"), "-o", exe]
        build = subprocess.run(cmd, capture_output=True, text=True)
        if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
            pytest.skip("sanitizer runtime not installed")
        assert build.returncode == 0, build.stderr
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
        assert run.returncode == 0 and "oracle sanitize run ok" in run.stdout, run.stdout + run.stderr


def test_product_host_code_under_asan_ubsan():
    """The product's host-side constant construction (table, RNS and gadget-basis builders in
    primus-fhe_amd/csrc/*.cpp) under ASan + UBSan; needs ROCm's clang (hipcc's host compiler)."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang not available")
    root = os.path.dirname(ORACLE)
    csrc = os.path.join(root, "primus-fhe_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_san")
        cmd = [clang, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
               "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", csrc,
               os.path.join(root, "tests", "support", "host_sanitize_main.cpp"),
               os.path.join(csrc, "pfhe_hosttables.cpp"), os.path.join(csrc, "pfhe_rns_host.cpp"), "-o", exe]
        build = subprocess.run(cmd, capture_output=True, text=True)
        if build.returncode != 0 and "asan" in build.stderr.lower() and "no such file" in build.stderr.lower():
            pytest.skip("sanitizer runtime not installed for ROCm clang")
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and "host sanitize run ok" in run.stdout, run.stdout + run.stderr
