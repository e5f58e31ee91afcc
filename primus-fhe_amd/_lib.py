"""ctypes loader for libpfhe_hip.so (the C ABI declared in include/pfhe.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFHE_LIB_PATH selects another build of the same library (tuning / ablation builds); default in-tree
_LIB_PATH = os.environ.get("PFHE_LIB_PATH") or os.path.join(_HERE, "libpfhe_hip.so")

u64p = C.POINTER(C.c_uint64)

STATUS = {
    0: "OK", 1: "NoPrimitiveRoot", 2: "DegreeConversionErr", 3: "DegreeTooLarge", 4: "NttTableErr",
    5: "ModulusTooLarge", 16: "EmptyBase", 17: "CoPrimeError", 18: "UnrepresentableModulus",
    32: "BadLength", 33: "BadArgument", 34: "NoDevice", 35: "HipError", 36: "Unsupported", 37: "NoInverse", 38: "Busy",
}


class PfheError(RuntimeError):
    """A non-zero pfhe_status returned by the C ABI."""

    def __init__(self, code: int, detail: str = ""):
        self.code = code
        self.kind = STATUS.get(code, f"status{code}")
        super().__init__(f"{self.kind} ({code}): {detail}" if detail else f"{self.kind} ({code})")


def library_path() -> str:
    return _LIB_PATH


def build(jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 into primus-fhe_amd/libpfhe_hip.so (in-tree)."""
    subprocess.run(["make", "-s", f"-j{jobs}", "-C", _HERE, "libpfhe_hip.so"], check=True)
    return _LIB_PATH


_lib = None


def _declare(lib: C.CDLL) -> None:
    vp, sz, u32, u64, ci = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("pfhe_status_string", C.c_char_p, ci)
    sig("pfhe_last_error", C.c_char_p)
    sig("pfhe_version", C.c_char_p)
    sig("pfhe_device_count", ci, C.POINTER(ci))
    sig("pfhe_device_malloc", ci, ci, sz, C.POINTER(vp))
    sig("pfhe_device_free", ci, ci, vp)
    sig("pfhe_memcpy_h2d", ci, ci, vp, vp, sz, vp)
    sig("pfhe_memcpy_d2h", ci, ci, vp, vp, sz, vp)
    sig("pfhe_memcpy_d2d", ci, ci, vp, vp, sz, vp)
    sig("pfhe_memset_dev", ci, ci, vp, ci, sz, vp)
    sig("pfhe_stream_copy_dev", ci, ci, vp, vp, sz, vp)
    sig("pfhe_stream_synchronize", ci, ci, vp)
    sig("pfhe_debug_alloc_count", u64)
    sig("pfhe_debug_stage_path_count", u64, ci)
    sig("pfhe_staging_release", ci, ci)
    sig("pfhe_fill_uniform_dev", ci, ci, vp, sz, u64p, sz, sz, u64, vp)

    sig("pfhe_ntt_create", ci, u32, u64, ci, C.POINTER(vp))
    sig("pfhe_ntt_destroy", None, vp)
    sig("pfhe_ntt_poly_length", sz, vp)
    sig("pfhe_ntt_log_n", u32, vp)
    for g in ("modulus", "root", "inv_root", "inv_n"):
        sig("pfhe_ntt_" + g, u64, vp)
    sig("pfhe_ntt_device", ci, vp)
    for g in ("transform_slice", "inverse_transform_slice", "lazy_transform_slice",
              "lazy_inverse_transform_slice"):
        sig("pfhe_ntt_" + g, ci, vp, vp, sz)
        sig("pfhe_dcrt_" + g, ci, vp, vp, sz)
    sig("pfhe_ntt_transform_monomial", ci, vp, u64, sz, vp, sz)
    sig("pfhe_ntt_transform_coeff_one_monomial", ci, vp, sz, vp, sz)
    sig("pfhe_ntt_transform_coeff_minus_one_monomial", ci, vp, sz, vp, sz)
    sig("pfhe_ntt_transform_dev", ci, vp, vp, sz, ci, vp)
    sig("pfhe_ntt_inverse_transform_dev", ci, vp, vp, sz, ci, vp)
    sig("pfhe_ntt_transform_monomial_dev", ci, vp, u64, sz, vp, sz, vp)
    sig("pfhe_ntt_mul_assign_dev", ci, vp, vp, sz, vp, sz, vp)
    sig("pfhe_ntt_add_mul_assign_dev", ci, vp, vp, vp, sz, vp, sz, vp)

    sig("pfhe_dcrt_create", ci, u32, u64p, sz, ci, C.POINTER(vp))
    sig("pfhe_dcrt_destroy", None, vp)
    for g in ("poly_length", "moduli_count", "crt_poly_length"):
        sig("pfhe_dcrt_" + g, sz, vp)
    sig("pfhe_dcrt_device", ci, vp)
    for g in ("modulus", "root", "inv_n"):
        sig("pfhe_dcrt_" + g, u64, vp, sz)
    sig("pfhe_dcrt_transform_monomial", ci, vp, u64, sz, vp, sz)
    sig("pfhe_dcrt_transform_coeff_one_monomial", ci, vp, sz, vp, sz)
    sig("pfhe_dcrt_transform_coeff_minus_one_monomial", ci, vp, sz, vp, sz)
    sig("pfhe_dcrt_transform_monomial_dev", ci, vp, u64, sz, vp, sz, ci, vp)
    sig("pfhe_dcrt_transform_dev", ci, vp, vp, sz, ci, vp)
    sig("pfhe_dcrt_inverse_transform_dev", ci, vp, vp, sz, ci, vp)
    sig("pfhe_dcrt_mul_assign_dev", ci, vp, vp, sz, vp, sz, vp)
    sig("pfhe_dcrt_add_mul_assign_dev", ci, vp, vp, vp, sz, vp, sz, vp)
    sig("pfhe_dcrt_mul_dcrt_polynomial_dev", ci, vp, vp, sz, vp, sz, vp)
    sig("pfhe_dcrt_butterfly_mul_dcrt_polynomial_to_dev", ci, vp, vp, vp, sz, vp, sz, vp, vp)
    sig("pfhe_dcrt_add_to_dev", ci, vp, vp, vp, vp, sz, vp)
    sig("pfhe_dcrt_sub_to_dev", ci, vp, vp, vp, vp, sz, vp)
    sig("pfhe_dcrt_neg_to_dev", ci, vp, vp, vp, sz, vp)
    sig("pfhe_dcrt_mul_scalar_to_dev", ci, vp, vp, u64p, vp, sz, vp)
    sig("pfhe_dcrt_add_mul_scalar_assign_dev", ci, vp, vp, vp, u64p, sz, vp)
    sig("pfhe_dcrt_mul_factor_to_dev", ci, vp, vp, u64p, vp, sz, vp)
    sig("pfhe_dcrt_add_mul_factor_assign_dev", ci, vp, vp, vp, u64p, sz, vp)
    sig("pfhe_dcrt_mul_monomial_to_dev", ci, vp, vp, sz, vp, sz, vp)
    sig("pfhe_dcrt_mul_monomial_assign_dev", ci, vp, vp, sz, sz, vp)
    sig("pfhe_dcrt_inv_to_dev", ci, vp, vp, vp, sz, vp)
    sig("pfhe_dcrt_butterfly_mul_factor_to_dev", ci, vp, vp, vp, sz, vp, sz, vp, vp)
    u8p = C.POINTER(C.c_uint8)
    # RNSBase<W> / BigUintApproxSignedBasis<W>: the same entry points for W = u64 (pfhe_rns, pfhe_basis) and
    # W = u32 (pfhe_rns32, pfhe_basis32)
    u32p = C.POINTER(u32)
    for rns, basis, w, wp in (("pfhe_rns_", "pfhe_basis_", u64, u64p), ("pfhe_rns32_", "pfhe_basis32_", u32, u32p)):
        sig(rns + "create", ci, wp, sz, ci, C.POINTER(vp))
        sig(rns + "destroy", None, vp)
        sig(rns + "moduli_count", sz, vp)
        sig(rns + "big_uint_value_len", sz, vp)
        sig(rns + "moduli_product", ci, vp, vp, sz)
        sig(rns + "compose_multiple_values_to", ci, vp, vp, sz, vp, sz, sz)
        sig(rns + "compose_multiple_values_to_dev", ci, vp, vp, sz, vp, sz, sz, vp)
        sig(rns + "wrapping_decompose_small_values_to", ci, vp, vp, sz, vp, sz, w)
        sig(rns + "wrapping_decompose_small_values_to_dev", ci, vp, vp, sz, vp, sz, w, vp)
        sig(rns + "add_wrapping_decompose_small_values_scaled", ci, vp, vp, sz, vp, sz, w, wp)
        sig(rns + "add_wrapping_decompose_small_values_scaled_dev", ci, vp, vp, sz, vp, sz, w, wp, vp)
        sig(rns + "add_decompose_small_values_scaled", ci, vp, vp, sz, vp, sz, wp)
        sig(rns + "add_decompose_small_values_scaled_dev", ci, vp, vp, sz, vp, sz, wp, vp)
        sig(rns + "decompose_big_uint_values_to", ci, vp, vp, sz, vp, sz, sz)
        sig(rns + "decompose_big_uint_values_to_dev", ci, vp, vp, sz, vp, sz, sz, vp)
        sig(basis + "create", ci, vp, u32, sz, C.POINTER(vp))
        sig(basis + "destroy", None, vp)
        sig(basis + "decompose_length", sz, vp)
        sig(basis + "log_basis", u32, vp)
        sig(basis + "drop_bits", u32, vp)
        sig(basis + "basis_value", w, vp)
        sig(basis + "scalars", ci, vp, vp, sz)
        sig(basis + "scalars_residue", ci, vp, vp, sz)
        sig(basis + "init_value_carry_slice_inplace", ci, vp, vp, sz, vp, sz)
        sig(basis + "init_value_carry_slice_inplace_dev", ci, vp, vp, sz, vp, sz, vp)
        sig(basis + "unsigned_decompose_slice_to", ci, vp, sz, vp, sz, vp, vp, sz)
        sig(basis + "unsigned_decompose_slice_to_dev", ci, vp, sz, vp, sz, vp, vp, sz, vp)
        sig(basis + "init_value_carry_slice_to", ci, vp, vp, sz, vp, vp, sz)
        sig(basis + "init_value_carry_slice_to_dev", ci, vp, vp, sz, vp, vp, sz, vp)
        sig(basis + "decompose_slice_to", ci, vp, sz, vp, sz, vp, sz, vp, sz)
        sig(basis + "decompose_slice_to_dev", ci, vp, sz, vp, sz, vp, sz, vp, sz, vp)
    for ep in ("pfhe_extprod_", "pfhe_extprod32_"):
        sig(ep + "plan_create", ci, vp, vp, vp, sz, sz, C.POINTER(vp))
        sig(ep + "plan_destroy", None, vp)
        sig(ep + "plan_scratch_bytes", sz, vp)
        sig(ep + "plan_in_use", ci, vp)
        sig(ep + "mul_dcrt_ggsw_to", ci, vp, vp, sz, vp, sz, vp, sz, ci)
        sig(ep + "mul_dcrt_ggsw_to_dev", ci, vp, vp, sz, vp, sz, vp, sz, ci, vp)
        sig(ep + "add_dcrt_glev_mul_crt_poly_assign_dev", ci, vp, vp, sz, vp, sz, vp, sz, vp)
        sig(ep + "glev_mul_crt_poly_to_dev", ci, vp, vp, sz, vp, sz, vp, sz, vp)
        sig(ep + "add_dcrt_glev_mul_big_uint_poly_assign_dev", ci, vp, vp, sz, vp, sz, vp, sz, vp)
        sig(ep + "glev_mul_big_uint_poly_to_dev", ci, vp, vp, sz, vp, sz, vp, sz, vp)
    sig("pfhe_extprod_plan_debug_hold", ci, vp, ci)
    sig("pfhe_extprod_profile_dev", ci, vp, vp, sz, vp, sz, vp, sz, C.POINTER(C.c_double), C.POINTER(sz), vp)
    sig("pfhe_dcrt_transform_num_passes", ci, vp)
    sig("pfhe_dcrt_transform_pass_name", C.c_char_p, vp, ci, ci)
    sig("pfhe_dcrt_transform_pass_dev", ci, vp, vp, sz, ci, ci, ci, vp)
    sig("pfhe_dcrt_transform_form", ci, vp, sz, ci, C.c_char_p, sz, C.POINTER(ci))

    for pre in ("pfhe_ntt_", "pfhe_dcrt_"):
        sig(pre + "mul_to_dev", ci, vp, vp, sz, vp, sz, vp, vp)
        sig(pre + "mul_add_to_dev", ci, vp, vp, sz, vp, sz, vp, vp, vp)
    sig("pfhe_dcrt_add_dcrt_glwe_mul_dcrt_polynomial_assign_dev", ci, vp, vp, vp, sz, vp, sz, sz, vp)
    sig("pfhe_dcrt_glwe_mul_dcrt_polynomial_to_dev", ci, vp, vp, sz, vp, sz, sz, vp, vp)
    for cv in ("pfhe_conv_", "pfhe_conv32_"):      # BaseConverter<u64> / BaseConverter<u32>
        sig(cv + "create", ci, vp, vp, C.POINTER(vp))
        sig(cv + "destroy", None, vp)
        sig(cv + "input_moduli_count", sz, vp)
        sig(cv + "output_moduli_count", sz, vp)
        sig(cv + "base_change_matrix", ci, vp, vp, sz)
        for g in ("fast_convert_array", "exact_convert_array"):
            sig(cv + g, ci, vp, vp, sz, vp, sz, sz)
            sig(cv + g + "_dev", ci, vp, vp, sz, vp, sz, sz, vp)
        sig(cv + "fast_convert_array_to_pairs_dev", ci, vp, vp, sz, vp, sz, sz, vp)

    # u32 tables
    sig("pfhe_ntt32_create", ci, u32, u32, ci, C.POINTER(vp))
    sig("pfhe_ntt32_destroy", None, vp)
    sig("pfhe_ntt32_poly_length", sz, vp)
    for g in ("log_n", "modulus", "root", "inv_root", "inv_n"):
        sig("pfhe_ntt32_" + g, u32, vp)
    sig("pfhe_ntt32_device", ci, vp)
    sig("pfhe_dcrt32_create", ci, u32, u32p, sz, ci, C.POINTER(vp))
    sig("pfhe_dcrt32_destroy", None, vp)
    for g in ("poly_length", "moduli_count", "crt_poly_length"):
        sig("pfhe_dcrt32_" + g, sz, vp)
    sig("pfhe_dcrt32_device", ci, vp)
    for g in ("modulus", "root"):
        sig("pfhe_dcrt32_" + g, u32, vp, sz)
    for pre in ("pfhe_ntt32_", "pfhe_dcrt32_"):
        for g in ("transform_slice", "inverse_transform_slice", "lazy_transform_slice",
                  "lazy_inverse_transform_slice"):
            sig(pre + g, ci, vp, vp, sz)
        sig(pre + "transform_monomial", ci, vp, u32, sz, vp, sz)
        sig(pre + "transform_coeff_one_monomial", ci, vp, sz, vp, sz)
        sig(pre + "transform_coeff_minus_one_monomial", ci, vp, sz, vp, sz)
        sig(pre + "transform_dev", ci, vp, vp, sz, ci, vp)
        sig(pre + "inverse_transform_dev", ci, vp, vp, sz, ci, vp)
        sig(pre + "mul_assign_dev", ci, vp, vp, sz, vp, sz, vp)
        sig(pre + "add_mul_assign_dev", ci, vp, vp, vp, sz, vp, sz, vp)
    sig("pfhe_ntt32_transform_monomial_dev", ci, vp, u32, sz, vp, sz, vp)
    sig("pfhe_dcrt32_fill_uniform_dev", ci, vp, vp, sz, u64, vp)
    sig("pfhe_dcrt32_transform_num_passes", ci, vp)
    sig("pfhe_dcrt32_transform_pass_name", C.c_char_p, vp, ci, ci)
    sig("pfhe_dcrt32_transform_pass_dev", ci, vp, vp, sz, ci, ci, ci, vp)
    sig("pfhe_dcrt32_transform_form", ci, vp, sz, ci, C.c_char_p, sz, C.POINTER(ci))


def lib() -> C.CDLL:
    """Load libpfhe_hip.so.  Fails loudly when it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  primus_fhe_amd has no CPU fallback.")
        try:
            # torch ships its own libamdhip64.so.7; load it first when torch is importable so
            # that one HIP runtime serves both torch (device memory, streams) and this library.
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        l = C.CDLL(_LIB_PATH)
        _declare(l)
        _lib = l
    return _lib


def status_string(code: int) -> str:
    return lib().pfhe_status_string(code).decode()


def check(code: int) -> None:
    if code != 0:
        raise PfheError(code, lib().pfhe_last_error().decode(errors="replace"))
