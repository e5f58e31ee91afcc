"""The C++ mirror (include/pfhe.hpp) compiles against the C ABI and links to libpfhe_hip.so;
without a GPU its constructors report NoDevice through pfhe::Error (no CPU fallback)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include "pfhe.hpp"
int main() {
    try {
        pfhe::U64DcrtTable t(10, {2305843009211596801ull, 2305843009210023937ull, 2305843009208713217ull});
        std::vector<uint64_t> a(3 * 1024, 1), b = a;
        t.transform_slice(a.data(), a.size());
        t.inverse_transform_slice(a.data(), a.size());
        pfhe::U32DcrtTable t32(10, {1073479681u, 1071513601u});
        std::vector<uint32_t> c(2 * 1024, 7), d = c;
        t32.transform_slice(c.data(), c.size());
        t32.inverse_transform_slice(c.data(), c.size());
        pfhe::RNSBase in({17, 19, 23}), out2({29, 31});
        pfhe::BaseConverter conv(in, out2);
        std::vector<uint64_t> res = {1, 2, 3}, conv_out(2);   // one coefficient, residues (1,2,3)
        conv.fast_convert_array(res.data(), 3, conv_out.data(), 2, 1);
        // the <u32> operators: compose(decompose(v)) == v over a nine-modulus base (constants in a device table),
        // and an external product with a zero GGSW is zero
        pfhe::RNSBase32 wide({1073707009u, 1073698817u, 1073692673u, 1073682433u, 1073668097u, 1073655809u, 1073651713u,
                              1073643521u, 1073620993u});
        const size_t wl = wide.big_uint_value_len();
        std::vector<uint32_t> big(4 * wl, 0), res32(9 * 4), back(4 * wl, 1);
        for (size_t i = 0; i < 4; ++i) { big[i * wl] = 12345u + (uint32_t)i; big[i * wl + 1] = 99u * (uint32_t)i; }
        wide.decompose_big_uint_values_to(big.data(), big.size(), res32.data(), res32.size(), 4);
        wide.compose_multiple_values_to(res32.data(), res32.size(), back.data(), back.size(), 4);
        pfhe::RNSBase32 in32({17u, 19u, 23u}), out32b({29u, 31u});
        pfhe::BaseConverter32 conv32(in32, out32b);
        std::vector<uint32_t> r32 = {1, 2, 3}, c32(2);
        conv32.fast_convert_array(r32.data(), 3, c32.data(), 2, 1);
        const bool conv32_ok = conv32.output_moduli_count() == 2 && c32[0] < 29 && c32[1] < 31;
        pfhe::RNSBase32 base32({1073479681u, 1071513601u});
        pfhe::BigUintApproxSignedBasis32 basis32(base32, 20);
        pfhe::DcrtGlevContext32 ctx32(t32, base32, basis32);
        const size_t ell32 = basis32.decompose_length(), cl = t32.crt_poly_length();
        std::vector<uint32_t> glwe32(2 * cl, 5), ggsw32(2 * ell32 * 2 * cl, 0), out32(2 * cl, 9);
        pfhe::mul_dcrt_ggsw_to(glwe32.data(), glwe32.size(), ggsw32.data(), ggsw32.size(), out32.data(), out32.size(), ctx32);
        bool u32_ok = conv32_ok && big == back && wide.moduli_count() == 9 && !ctx32.in_use();
        for (uint32_t w : out32) u32_ok = u32_ok && w == 0;
        // element-wise family on device buffers: ((a + b) - b) * X^5 * X^(2N-5) == a, and -(-a) == a
        void *da = nullptr, *db = nullptr, *dx = nullptr;
        const size_t bytes = a.size() * sizeof(uint64_t);
        pfhe::check(pfhe_device_malloc(0, bytes, &da));
        pfhe::check(pfhe_device_malloc(0, bytes, &db));
        pfhe::check(pfhe_device_malloc(0, bytes, &dx));
        std::vector<uint64_t> e(a.size());
        for (size_t i = 0; i < e.size(); ++i) e[i] = 1000003ull * i + 17;
        pfhe::check(pfhe_memcpy_h2d(0, da, e.data(), bytes, nullptr));
        pfhe::check(pfhe_memcpy_h2d(0, db, b.data(), bytes, nullptr));
        uint64_t *pa = (uint64_t *)da, *pb = (uint64_t *)db, *px = (uint64_t *)dx;
        t.add_to_dev(pa, pb, px, e.size());
        t.sub_to_dev(px, pb, px, e.size());
        t.mul_monomial_assign_dev(px, 5, e.size());
        t.mul_monomial_to_dev(px, 2 * 1024 - 5, pb, e.size());
        t.neg_to_dev(pb, pb, e.size());
        t.neg_to_dev(pb, pb, e.size());
        t.mul_scalar_to_dev(pb, {2, 2, 2}, px, e.size());
        std::vector<uint64_t> f(e.size());
        pfhe::check(pfhe_memcpy_d2h(0, f.data(), dx, bytes, nullptr));
        bool ew_ok = true;
        for (size_t i = 0; i < e.size(); ++i) ew_ok = ew_ok && f[i] == 2 * e[i];
        pfhe_device_free(0, da); pfhe_device_free(0, db); pfhe_device_free(0, dx);
        const bool ok = ew_ok && u32_ok && a == b && c == d && conv.input_moduli_count() == 3 && conv_out[0] < 29 && conv_out[1] < 31;
        std::printf(ok ? "roundtrip ok\n" : "roundtrip MISMATCH\n");
        return ok ? 0 : 1;
    } catch (const pfhe::Error &e) {
        std::printf("pfhe::Error %d: %s\n", e.status(), e.what());
        return e.status() == PFHE_ERR_NO_DEVICE ? 42 : 2;
    }
}
'''


def test_cpp_mirror_compiles_and_runs():
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    import primus_fhe_amd as p
    lib_dir = os.path.dirname(p.library_path())
    if not os.path.exists(p.library_path()):
        pytest.skip("libpfhe_hip.so not built")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", lib_dir, "-lpfhe_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        import torch
        if torch.cuda.is_available():
            assert r.returncode == 0 and "roundtrip ok" in r.stdout, r.stdout + r.stderr
        else:
            assert r.returncode == 42 and "NO_DEVICE" not in r.stderr, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    """The same program on the GPU box: transforms, base conversion and the element-wise family round-trip."""
    test_cpp_mirror_compiles_and_runs()
