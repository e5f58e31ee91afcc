#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.json).

The reference is Rust and cannot be built in this image, so these vectors are NOT
outputs of the reference binary.  Each expected value is produced by the pure-Python big-integer
implementation in tests/pyref.py (textbook definitions on Python ints) and, before it is written,
compared with the C restatement in oracle/ -- the generator refuses to write a fixture the two
disagree on.  Large-size entries are SHA-256 digests of the little-endian u64 output words.

Inputs come from a SplitMix64 stream seeded 0x5EED_0000_0000_0000 + case id, drawn by rejection to
uniform [0, q), so tests regenerate them instead of storing them.

Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import pyref  # noqa: E402
from golden_inputs import digest, splitmix_rns, splitmix_uniform  # noqa: E402
from oracle import oracle as orc  # noqa: E402

Q61, Q62 = pyref.Q61, pyref.Q62


def small_ntt_cases():
    cases = []
    cid = 0
    for log_n, q in [(1, 17), (2, 17), (3, 97), (3, Q62), (4, 132120577), (5, 1125899906826241),
                     (6, Q61[0]), (6, Q61[2]), (6, 1152921504606830593)]:
        n = 1 << log_n
        a = splitmix_uniform(0x100 + cid, q, n).tolist()
        b = splitmix_uniform(0x200 + cid, q, n).tolist()
        psi = pyref.minimal_primitive_root(log_n + 1, q)
        fwd = pyref.ntt_direct(a, q, log_n, psi).tolist()
        prod = pyref.negacyclic_mul(a, b, q)
        # agree with the C restatement before writing
        t = orc.U64NttTable(log_n, q)
        assert t.root == psi, (log_n, q)
        x = np.array(a, np.uint64)
        t.transform_slice(x)
        assert x.tolist() == fwd, (log_n, q)
        t.inverse_transform_slice(x)
        assert x.tolist() == a
        assert orc.naive_negacyclic_mul(q, np.array(a, np.uint64), np.array(b, np.uint64)).tolist() == prod
        cases.append(dict(case=cid, log_n=log_n, q=q, root=psi, seed_a=0x100 + cid, seed_b=0x200 + cid,
                          a=[str(v) for v in a], b=[str(v) for v in b],
                          ntt_a=[str(v) for v in fwd], a_mul_b=[str(v) for v in prod]))
        cid += 1
    return cases


def rns_gadget_cases():
    out = []
    for cid, (moduli, log_basis, rev) in enumerate([(Q61, 30, None), (Q61[:2], 20, 3), ([97, 101, 103], 4, None),
                                                    ([1125899906826241, 1125899906629633], 7, None)]):
        count = 12
        res = splitmix_rns(0x300 + cid, moduli, count)  # modulus-major: [q0 x count][q1 x count]...
        L = len(moduli)
        g = pyref.Gadget(moduli, log_basis, rev)
        vals = [pyref.crt_compose([int(res[r * count + c]) for r in range(L)], moduli) for c in range(count)]
        digits = [g.signed_digits(v) for v in vals]
        # oracle agreement
        ob = orc.RNSBase(moduli)
        ov = ob.compose_multiple_values_to(res, count)
        W = ob.value_len
        assert [pyref.limbs_to_int(ov[c * W:(c + 1) * W]) for c in range(count)] == vals
        obasis = orc.BigUintApproxSignedBasis(ob, log_basis, rev)
        assert obasis.decompose_length == g.ell and obasis.drop_bits == g.drop
        carries = obasis.init_value_carry_slice_inplace(ov, count)
        for j in range(g.ell):
            u = obasis.unsigned_decompose_slice_to(j, ov, carries, count)
            assert [int(x) for x in u] == [g.unsigned_digits(v)[j] for v in vals], (cid, j)
        out.append(dict(case=cid, moduli=[str(m) for m in moduli], log_basis=log_basis, reverse_length=rev,
                        decompose_length=g.ell, drop_bits=g.drop, seed=0x300 + cid, count=count,
                        values=[hex(v) for v in vals], signed_digits=[[str(d) for d in ds] for ds in digits]))
    return out


def small_extprod_case():
    log_n, k, moduli, log_basis = 3, 1, Q61, 30
    n, L = 1 << log_n, 3
    g = pyref.Gadget(moduli, log_basis)
    glwe = splitmix_rns(0x400, moduli, n, k + 1)
    key = splitmix_rns(0x401, moduli, n, (k + 1) * g.ell * (k + 1))
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, L, n).tolist(),
                                       key.reshape(k + 1, g.ell, k + 1, L, n).tolist())
    ot, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis)
    ggsw = key.copy()
    ot.transform_slice(ggsw)
    o = orc.mul_dcrt_ggsw_to(ot, ob, obasis, k, glwe, ggsw)
    ot.inverse_transform_slice(o)
    assert o.reshape(k + 1, L, n).tolist() == exp
    return dict(log_n=log_n, k=k, moduli=[str(m) for m in moduli], log_basis=log_basis, seed_glwe=0x400,
                seed_key_coeff=0x401, result_coeff=[str(v) for v in np.array(exp, dtype=object).reshape(-1)])


def digests():
    d = []
    # single-prime forward NTT + round trip, fast big-int Python NTT as the second implementation
    for cid, (log_n, q, batch) in enumerate([(10, Q62, 2), (12, 1125899906826241, 2), (14, Q61[0], 2),
                                             (16, Q61[0], 1), (16, Q61[1], 1), (16, Q61[2], 1)]):
        n = 1 << log_n
        a = splitmix_uniform(0x500 + cid, q, n * batch)
        psi = pyref.minimal_primitive_root(log_n + 1, q)
        py = np.concatenate([np.array(pyref.ntt_fast(a[i * n:(i + 1) * n].tolist(), q, log_n, psi), np.uint64)
                             for i in range(batch)])
        t = orc.U64NttTable(log_n, q)
        x = a.copy()
        t.transform_slice(x)
        assert np.array_equal(x, py), (log_n, q)
        d.append(dict(kind="ntt_forward", case=cid, log_n=log_n, q=str(q), batch=batch, seed=0x500 + cid,
                      root=str(psi), input_sha256=digest(a), output_sha256=digest(x)))
    # RNS (DCRT) polynomial product, config-3 shape
    for cid, (log_n, moduli, batch) in enumerate([(10, Q61, 2), (16, Q61, 1)]):
        n = 1 << log_n
        a = splitmix_rns(0x600 + cid, moduli, n, batch)
        b = splitmix_rns(0x610 + cid, moduli, n, batch)
        t = orc.U64DcrtTable(log_n, moduli)
        fa, fb = a.copy(), b.copy()
        t.transform_slice(fa)
        t.transform_slice(fb)
        W = 3 * n
        for e in range(batch):
            t.mul_assign(fa[e * W:(e + 1) * W], fb[e * W:(e + 1) * W])
        t.inverse_transform_slice(fa)
        if log_n <= 10:  # schoolbook on Python ints
            for e in range(batch):
                for r, q in enumerate(moduli):
                    s = slice((e * 3 + r) * n, (e * 3 + r + 1) * n)
                    assert fa[s].tolist() == pyref.negacyclic_mul(a[s].tolist(), b[s].tolist(), q)
        else:  # fast big-int NTT product
            for r, q in enumerate(moduli):
                s = slice(r * n, (r + 1) * n)
                assert fa[s].tolist() == pyref.negacyclic_mul_fast(a[s].tolist(), b[s].tolist(), q, log_n)
        d.append(dict(kind="dcrt_polymul", case=cid, log_n=log_n, moduli=[str(m) for m in moduli], batch=batch,
                      seed_a=0x600 + cid, seed_b=0x610 + cid, output_sha256=digest(fa)))
    # external product, config-4 shape at batch 1 and a mid-size case (oracle; pinned at N=8 by
    # small_extprod_case and by tests/test_oracle_rns_gadget.py against pyref)
    for cid, (log_n, k, moduli, log_basis, batch) in enumerate([(10, 1, Q61, 30, 2), (16, 1, Q61, 30, 1)]):
        n = 1 << log_n
        ot, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
        obasis = orc.BigUintApproxSignedBasis(ob, log_basis)
        ell = obasis.decompose_length
        glwe = splitmix_rns(0x700 + cid, moduli, n, batch * (k + 1))
        ggsw = splitmix_rns(0x710 + cid, moduli, n, (k + 1) * ell * (k + 1))  # NTT-domain key, shared
        W = (k + 1) * 3 * n
        res = np.concatenate([orc.mul_dcrt_ggsw_to(ot, ob, obasis, k, glwe[e * W:(e + 1) * W].copy(), ggsw)
                              for e in range(batch)])
        d.append(dict(kind="external_product", case=cid, log_n=log_n, k=k, moduli=[str(m) for m in moduli],
                      log_basis=log_basis, batch=batch, seed_glwe=0x700 + cid, seed_ggsw=0x710 + cid,
                      output_sha256=digest(res)))
    return d


def u32_cases():
    """U32NttTable (prime32): full vectors at N <= 64 from the big-integer evaluation, digests at 2^10 / 2^16."""
    small, dig = [], []
    for cid, (log_n, q) in enumerate([(1, 17), (3, 132120577), (5, 268369921), (6, 1073479681)]):
        n = 1 << log_n
        a = splitmix_uniform(0x800 + cid, q, n).tolist()
        psi = pyref.minimal_primitive_root(log_n + 1, q)
        fwd = pyref.ntt_direct(a, q, log_n, psi).tolist()
        t = orc.U32NttTable(log_n, q)
        assert t.root == psi
        x = np.array(a, np.uint32)
        t.transform_slice(x)
        assert x.tolist() == fwd
        small.append(dict(case=cid, log_n=log_n, q=q, root=psi, seed=0x800 + cid, a=a, ntt_a=fwd))
    for cid, (log_n, q, batch) in enumerate([(10, 132120577, 2), (16, 1073479681, 1)]):
        n = 1 << log_n
        a = splitmix_uniform(0x810 + cid, q, n * batch)
        t = orc.U32NttTable(log_n, q)
        x = a.astype(np.uint32)
        t.transform_slice(x)
        py = np.concatenate([np.array(pyref.ntt_fast(a[i * n:(i + 1) * n].tolist(), q, log_n), np.uint64)
                             for i in range(batch)])
        assert np.array_equal(x.astype(np.uint64), py)
        dig.append(dict(case=cid, log_n=log_n, q=q, batch=batch, seed=0x810 + cid,
                        output_sha256=hashlib.sha256(np.ascontiguousarray(x, dtype="<u4").tobytes()).hexdigest()))
    return dict(small=small, digests=dig)


def converter_cases():
    """BaseConverter: fast_convert_array by its definition on Python integers; exact_convert_array as the
    centred representative (inputs kept away from Q/2 so that no float rounding decides the branch)."""
    out = []
    for cid, (mod_in, mod_out) in enumerate([([17, 19, 23], [29, 31]), (Q61, [1152921504606584833, 1125899906826241]),
                                             (Q61[:2], Q61[2:])]):
        n = 16
        Q = 1
        for q in mod_in:
            Q *= q
        x = np.concatenate([splitmix_uniform(0x900 + 16 * cid + i, q, n) for i, q in enumerate(mod_in)])
        vals = [pyref.crt_compose([int(x[i * n + t]) for i in range(len(mod_in))], mod_in) for t in range(n)]
        fast = []
        for j, p in enumerate(mod_out):
            for t in range(n):
                s_ = sum((int(x[i * n + t]) * pow(Q // q, -1, q) % q) * (Q // q) for i, q in enumerate(mod_in))
                fast.append(s_ % p)
        oin = orc.RNSBase(mod_in)
        conv = orc.BaseConverter(oin, orc.RNSBase(mod_out))
        assert conv.fast_convert_array(x, n).tolist() == fast
        p0 = mod_out[0]
        exact = [(v % p0 if 2 * v < Q else (v - Q) % p0) for v in vals]
        assert all(abs(v / Q - 0.5) > 1e-6 for v in vals)
        econv = orc.BaseConverter(oin, orc.RNSBase([p0]))
        assert econv.exact_convert_array(x, n).tolist() == exact
        out.append(dict(case=cid, input_moduli=[str(m) for m in mod_in], output_moduli=[str(m) for m in mod_out],
                        seed_base=0x900 + 16 * cid, count=n, fast=[str(v) for v in fast],
                        exact_to_first_output_modulus=[str(v) for v in exact]))
    return out


def _elementwise_expected(moduli, n, a, b, scalars, r):
    """Every op of the element-wise family by its definition on Python integers (canonical residues)."""
    L = len(moduli)
    unit = L * n
    A, B = [int(v) for v in a], [int(v) for v in b]
    q_of = lambda i: moduli[(i // n) % L]
    s_of = lambda i: scalars[(i // n) % L]
    exp = {
        "add": [(x + y) % q_of(i) for i, (x, y) in enumerate(zip(A, B))],
        "sub": [(x - y) % q_of(i) for i, (x, y) in enumerate(zip(A, B))],
        "neg": [(-x) % q_of(i) for i, x in enumerate(A)],
        "mul_scalar": [x * s_of(i) % q_of(i) for i, x in enumerate(A)],
        "add_mul_scalar": [(x + y * s_of(i)) % q_of(i) for i, (x, y) in enumerate(zip(A, B))],
        "inv": [pow(x, -1, q_of(i)) for i, x in enumerate(A)],
    }
    mono = [0] * len(A)
    for i, x in enumerate(A):  # a * X^r in Z_q[X]/(X^n + 1), per n-word polynomial
        base, j = (i // n) * n, i % n
        d = j + r
        sign = -1 if (d // n) % 2 else 1
        mono[base + d % n] = (sign * x) % q_of(i)
    exp["mul_monomial"] = mono
    assert len(A) % unit == 0
    return exp


def elementwise_cases():
    """add / sub / neg / mul_scalar / add_mul_scalar / mul_factor / add_mul_factor / mul_monomial / inv
    (primus_poly crt/{add,sub,neg,mul}.rs, dcrt/inv.rs): full vectors at N = 8, digests at N = 2^12."""
    out = {"small": [], "digests": []}
    for cid, (log_n, moduli, batch, r) in enumerate([(3, [97, Q61[0]], 2, 3), (3, [Q62], 1, 13), (12, Q61, 4, 1000),
                                                      (12, Q61, 4, 4096 + 77), (10, [1125899906826241], 3, 0)]):
        n = 1 << log_n
        a = splitmix_rns(0xA00 + 2 * cid, moduli, n, batch)
        b = splitmix_rns(0xA01 + 2 * cid, moduli, n, batch)
        a[a == 0] = 1  # inv needs units
        scalars = [int(splitmix_uniform(0xA80 + cid, q, 1)[0]) for q in moduli]
        factors = [v for s_, q in zip(scalars, moduli) for v in (s_, (s_ << 64) // q)]
        exp = _elementwise_expected(moduli, n, a, b, scalars, r)
        exp["mul_factor"], exp["add_mul_factor"] = exp["mul_scalar"], exp["add_mul_scalar"]
        o = orc.CrtPolyOps(moduli, n)
        got = {"add": o.add_to(a, b), "sub": o.sub_to(a, b), "neg": o.neg_to(a), "mul_scalar": o.mul_scalar_to(a, scalars),
               "mul_factor": o.mul_factor_to(a, factors), "inv": o.inv_to(a)}
        acc = a.copy(); o.add_mul_scalar_assign(acc, b, scalars); got["add_mul_scalar"] = acc
        acc = a.copy(); o.add_mul_factor_assign(acc, b, factors); got["add_mul_factor"] = acc
        m = a.copy(); o.mul_monomial_assign(m, r); got["mul_monomial"] = m
        for k_, v in exp.items():
            assert got[k_].tolist() == v, (cid, k_)
        head = dict(case=cid, log_n=log_n, moduli=[str(q) for q in moduli], batch=batch, r=r, seed_a=0xA00 + 2 * cid,
                    seed_b=0xA01 + 2 * cid, scalars=[str(v) for v in scalars], factors=[str(v) for v in factors])
        if log_n <= 3:
            out["small"].append(dict(head, a=[str(v) for v in a], b=[str(v) for v in b],
                                     expected={k_: [str(x) for x in v] for k_, v in exp.items()}))
        else:
            out["digests"].append(dict(head, sha256={k_: digest(np.array(v, np.uint64)) for k_, v in exp.items()}))
    return out


def main():
    out = {
        "u32_ntt.json": u32_cases(),
        "base_converter.json": converter_cases(),
        "elementwise.json": elementwise_cases(),
        "ntt_small.json": small_ntt_cases(),
        "rns_gadget_small.json": rns_gadget_cases(),
        "extprod_small.json": small_extprod_case(),
        "digests.json": digests(),
    }
    for name, obj in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1)
            f.write("\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
