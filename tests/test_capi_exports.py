"""CPU-side boundary checks: libpfhe_hip.so builds for gfx950, loads without a GPU, exports every
symbol include/pfhe.h declares, and fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pfhe():
    import primus_fhe_amd as p
    if not os.path.exists(p.library_path()):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("hipcc not available to build libpfhe_hip.so")
        p.build()
    return p


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pfhe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfhe_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pfhe):
    lib = ctypes.CDLL(pfhe.library_path())
    names = declared_functions()
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in pfhe.h but not exported: {missing}"


def test_status_strings(pfhe):
    assert pfhe.status_string(0) == "ok"
    assert "primitive root" in pfhe.status_string(1)
    assert b"gfx950" in pfhe.lib().pfhe_version()


def test_no_cpu_fallback_without_device(pfhe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.U64NttTable(10, 4611686018425815041)
    assert e.value.kind == "NoDevice"
    # argument errors are reported before the device is touched, mirroring NttTable::new
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.U64NttTable(20, 1125899906826241)   # 2N does not divide q-1
    assert e.value.kind == "NoPrimitiveRoot"
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.U64DcrtTable(4, [])
    assert e.value.kind == "BadArgument"


def test_rns_base_moduli_cap_is_reported_before_the_device_is_touched(pfhe):
    """RNSBase::new (primus_rns/src/base.rs:79-117) takes any number of moduli; here up to 8 travel as kernel arguments
    and up to 32 in a device table (csrc/pfhe_rns.hpp).  A 33rd modulus is refused with PFHE_ERR_UNSUPPORTED — not
    truncated, not a crash — and the reference's own errors keep their precedence.  (Nine moduli are accepted: on a box
    without a GPU the constructor gets as far as the device check.)"""
    from primes import ntt_primes_below
    primes = ntt_primes_below(33, 60, 16)
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.RNSBase(primes)
    assert e.value.kind == "Unsupported" and "32" in str(e.value)
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.RNSBase(primes[:32] + [primes[0]])      # CoPrimeError first (base.rs:83-89)
    assert e.value.kind == "CoPrimeError"
    try:
        assert pfhe.RNSBase(primes[:9]).moduli_count() == 9
    except pfhe.PfheError as err:
        assert err.kind == "NoDevice"
    with pytest.raises(pfhe.PfheError) as e:
        pfhe.RNSBase([])
    assert e.value.kind == "EmptyBase"
