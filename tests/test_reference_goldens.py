"""The pin the oracle is waiting for.

tests/golden/reference_digests.json is written by integration/emit_golden — a Rust binary that runs the REAL primus-fhe
crates (U64NttTable, U32NttTable, U64DcrtTable + DcrtPolynomial::mul_assign, RNSBase::compose_multiple_values_to,
BigUintApproxSignedBasis digits, CrtGlwe::mul_dcrt_ggsw_to) on this repository's SplitMix64 input streams.  No Rust
toolchain exists in the build image, so the file is ABSENT today and the two tests that consume it are skipped; the day a
maintainer with cargo runs the one command in INTEGRATION.md ("Pinning the oracle") and commits the JSON, the oracle
(CPU suite) and the HIP path (`-m gpu`) must both reproduce every digest in it, and "parity unpinned" comes out of
DESIGN.md.  Only the JSON travels: nothing of the reference is read at test time.

The consuming logic itself is exercised now, against a hand-made file in a temporary directory (written from the oracle
and marked as such): a faithful file passes, a tampered digest and an unknown kind are reported.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import pyref
from golden_inputs import digest, splitmix_rns, splitmix_uniform
from primes import ntt_primes_below

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_FILE = os.path.join(HERE, "golden", "reference_digests.json")
Q61 = pyref.Q61
W9 = ntt_primes_below(9, 61, 4)     # the nine largest primes below 2^61 that are 1 mod 32 (emit_golden: const W9)
Q60 = ntt_primes_below(3, 60, 4)    # conversion target (const Q60)
Q30 = [1073479681, 1071513601, 1070727169]
P27 = [134215681, 134176769]       # the moduli of primus_decompose/tests/big_uint.rs:21 (const P27)

# cases beyond tests/golden/digests.json that integration/emit_golden emits too (keep in step with its main.rs)
EXTRA_CASES = [
    dict(kind="ntt32_forward", case=0, log_n=10, q="132120577", batch=2, seed=0x810),
    dict(kind="ntt32_forward", case=1, log_n=16, q="1073479681", batch=1, seed=0x811),
    dict(kind="rns_compose", case=0, moduli=[str(q) for q in Q61], count=4096, seed=0x340),
    dict(kind="gadget_digits", case=0, moduli=[str(q) for q in Q61], log_basis=30, count=4096, seed=0x340),
    dict(kind="rns_compose", case=1, moduli=[str(q) for q in Q61], count=1000, seed=0x341),
    dict(kind="gadget_digits", case=1, moduli=[str(q) for q in Q61], log_basis=13, count=1000, seed=0x341),
    # round 6: a base of nine moduli (compose, digits, conversion into three 60-bit moduli) and the u32 external product
    dict(kind="rns_compose", case=2, moduli=[str(q) for q in W9], count=2048, seed=0x342),
    dict(kind="gadget_digits", case=2, moduli=[str(q) for q in W9], log_basis=30, count=2048, seed=0x342),
    dict(kind="base_convert", case=0, moduli=[str(q) for q in W9], moduli_out=[str(q) for q in Q60], count=2048, seed=0x342),
    dict(kind="external_product32", case=0, log_n=10, k=1, moduli=[str(q) for q in Q30], log_basis=15, batch=2,
         seed_glwe=0x720, seed_ggsw=0x730),
    dict(kind="base_convert32", case=0, moduli=[str(q) for q in Q30], moduli_out=[str(q) for q in P27], count=2048, seed=0x350),
]


def digest_u32(words) -> str:
    return hashlib.sha256(np.ascontiguousarray(words, dtype="<u4").tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------ the two backends
class OracleBackend:
    name = "oracle"

    def __init__(self, orc):
        self.o = orc

    def ntt_forward(self, log_n, q, x, n):
        t = self.o.U64NttTable(log_n, q)
        for i in range(0, x.size, n):
            t.transform_slice(x[i:i + n])
        return t.root, x

    def ntt32_forward(self, log_n, q, x, n):
        t = self.o.U32NttTable(log_n, q)
        for i in range(0, x.size, n):
            t.transform_slice(x[i:i + n])
        return x

    def dcrt_polymul(self, log_n, moduli, a, b):
        t = self.o.U64DcrtTable(log_n, moduli)
        W = t.crt_poly_length
        t.transform_slice(a)
        t.transform_slice(b)
        for e in range(a.size // W):
            t.mul_assign(a[e * W:(e + 1) * W], b[e * W:(e + 1) * W])
        t.inverse_transform_slice(a)
        return a

    def external_product(self, log_n, k, moduli, log_basis, glwe, ggsw_of):
        t, base = self.o.U64DcrtTable(log_n, moduli), self.o.RNSBase(moduli)
        basis = self.o.BigUintApproxSignedBasis(base, log_basis)
        ggsw = ggsw_of(basis.decompose_length)
        W = (k + 1) * t.crt_poly_length
        return np.concatenate([self.o.mul_dcrt_ggsw_to(t, base, basis, k, glwe[e * W:(e + 1) * W].copy(), ggsw)
                               for e in range(glwe.size // W)])

    def base_convert(self, moduli, moduli_out, residues, count):
        inp = self.o.RNSBase(moduli)
        fast = self.o.BaseConverter(inp, self.o.RNSBase(moduli_out)).fast_convert_array(residues, count)
        exact = self.o.BaseConverter(inp, self.o.RNSBase(moduli_out[:1])).exact_convert_array(residues, count)
        return np.concatenate([fast, exact])

    def base_convert32(self, moduli, moduli_out, residues, count):
        inp = self.o.RNSBase32(moduli)
        fast = self.o.BaseConverter32(inp, self.o.RNSBase32(moduli_out)).fast_convert_array(residues, count)
        exact = self.o.BaseConverter32(inp, self.o.RNSBase32(moduli_out[:1])).exact_convert_array(residues, count)
        return np.concatenate([fast, exact])

    def external_product32(self, log_n, k, moduli, log_basis, glwe, ggsw_of):
        t, base = self.o.U32DcrtTable(log_n, moduli), self.o.RNSBase32(moduli)
        basis = self.o.BigUintApproxSignedBasis32(base, log_basis)
        ggsw = ggsw_of(basis.decompose_length)
        W = (k + 1) * t.crt_poly_length
        return np.concatenate([self.o.mul_dcrt32_ggsw_to(t, base, basis, k, glwe[e * W:(e + 1) * W].copy(), ggsw)
                               for e in range(glwe.size // W)])

    def compose_and_digits(self, moduli, log_basis, residues, count):
        base = self.o.RNSBase(moduli)
        vals = base.compose_multiple_values_to(residues, count)
        composed = vals.copy()
        if log_basis is None:
            return composed, None, None
        basis = self.o.BigUintApproxSignedBasis(base, log_basis)
        carries = basis.init_value_carry_slice_inplace(vals, count)
        levels = [basis.unsigned_decompose_slice_to(j, vals, carries, count).copy() for j in range(basis.decompose_length)]
        return composed, np.concatenate(levels), (basis.decompose_length, basis.drop_bits)


class HipBackend:
    name = "hip"

    def __init__(self, pf):
        self.p = pf

    def ntt_forward(self, log_n, q, x, n):
        t = self.p.U64NttTable(log_n, q)
        t.transform_slice(x)
        return t.root(), x

    def ntt32_forward(self, log_n, q, x, n):
        t = self.p.U32NttTable(log_n, q)
        t.transform_slice(x)
        return x

    def dcrt_polymul(self, log_n, moduli, a, b):
        from gpu_util import to_dev, to_host
        t = self.p.U64DcrtTable(log_n, moduli)
        da, db = to_dev(a), to_dev(b)
        t.transform_dev(da)
        t.transform_dev(db)
        t.mul_assign_dev(da, db)
        t.inverse_transform_dev(da)
        return to_host(da)

    def external_product(self, log_n, k, moduli, log_basis, glwe, ggsw_of):
        t, base = self.p.U64DcrtTable(log_n, moduli), self.p.RNSBase(moduli)
        basis = self.p.BigUintApproxSignedBasis(base, log_basis)
        ctx = self.p.DcrtGlevContext(t, base, basis, k)
        out = np.empty_like(glwe)
        self.p.mul_dcrt_ggsw_to(glwe, ggsw_of(basis.decompose_length()), out, ctx)
        return out

    def base_convert(self, moduli, moduli_out, residues, count):
        inp = self.p.RNSBase(moduli)
        fast = np.empty(len(moduli_out) * count, np.uint64)
        self.p.BaseConverter(inp, self.p.RNSBase(moduli_out)).fast_convert_array(residues, fast, count)
        exact = np.empty(count, np.uint64)
        self.p.BaseConverter(inp, self.p.RNSBase(moduli_out[:1])).exact_convert_array(residues, exact, count)
        return np.concatenate([fast, exact])

    def base_convert32(self, moduli, moduli_out, residues, count):
        inp = self.p.RNSBase32(moduli)
        fast = np.empty(len(moduli_out) * count, np.uint32)
        self.p.BaseConverter32(inp, self.p.RNSBase32(moduli_out)).fast_convert_array(residues, fast, count)
        exact = np.empty(count, np.uint32)
        self.p.BaseConverter32(inp, self.p.RNSBase32(moduli_out[:1])).exact_convert_array(residues, exact, count)
        return np.concatenate([fast, exact])

    def external_product32(self, log_n, k, moduli, log_basis, glwe, ggsw_of):
        t, base = self.p.U32DcrtTable(log_n, moduli), self.p.RNSBase32(moduli)
        basis = self.p.BigUintApproxSignedBasis32(base, log_basis)
        ctx = self.p.DcrtGlevContext32(t, base, basis, k)
        out = np.empty_like(glwe)
        self.p.mul_dcrt_ggsw_to(glwe, ggsw_of(basis.decompose_length()), out, ctx)
        return out

    def compose_and_digits(self, moduli, log_basis, residues, count):
        base = self.p.RNSBase(moduli)
        W = base.big_uint_value_len()
        vals = np.empty(count * W, np.uint64)
        base.compose_multiple_values_to(residues, vals, count)
        composed = vals.copy()
        if log_basis is None:
            return composed, None, None
        basis = self.p.BigUintApproxSignedBasis(base, log_basis)
        carries = np.zeros(count, np.uint8)
        basis.init_value_carry_slice_inplace(vals, carries)
        levels = []
        for j in range(basis.decompose_length()):
            u = np.empty(count, np.uint64)
            basis.unsigned_decompose_slice_to(j, vals, u, carries)
            levels.append(u)
        return composed, np.concatenate(levels), (basis.decompose_length(), basis.drop_bits())


# ------------------------------------------------------------------------------------------ one entry -> digest
def compute(entry, be):
    """(digest of the backend's output for `entry`, extra fields the entry may pin) — inputs regenerated from the seeds."""
    kind = entry["kind"]
    if kind == "ntt_forward":
        q, n = int(entry["q"]), 1 << entry["log_n"]
        x = splitmix_uniform(entry["seed"], q, n * entry["batch"])
        extra = {"input_sha256": digest(x)}
        root, y = be.ntt_forward(entry["log_n"], q, x, n)
        extra["root"] = str(root)
        return digest(y), extra
    if kind == "ntt32_forward":
        q, n = int(entry["q"]), 1 << entry["log_n"]
        x = splitmix_uniform(entry["seed"], q, n * entry["batch"]).astype(np.uint32)
        return digest_u32(be.ntt32_forward(entry["log_n"], q, x, n)), {}
    moduli = [int(m) for m in entry["moduli"]]
    if kind == "dcrt_polymul":
        n = 1 << entry["log_n"]
        a = splitmix_rns(entry["seed_a"], moduli, n, entry["batch"])
        b = splitmix_rns(entry["seed_b"], moduli, n, entry["batch"])
        return digest(be.dcrt_polymul(entry["log_n"], moduli, a, b)), {}
    if kind == "external_product":
        n, k = 1 << entry["log_n"], entry["k"]
        glwe = splitmix_rns(entry["seed_glwe"], moduli, n, entry["batch"] * (k + 1))
        ggsw_of = lambda ell: splitmix_rns(entry["seed_ggsw"], moduli, n, (k + 1) * ell * (k + 1))  # noqa: E731
        return digest(be.external_product(entry["log_n"], k, moduli, entry["log_basis"], glwe, ggsw_of)), {}
    if kind == "external_product32":
        n, k = 1 << entry["log_n"], entry["k"]
        glwe = splitmix_rns(entry["seed_glwe"], moduli, n, entry["batch"] * (k + 1)).astype(np.uint32)
        ggsw_of = lambda ell: splitmix_rns(entry["seed_ggsw"], moduli, n, (k + 1) * ell * (k + 1)).astype(np.uint32)  # noqa: E731
        return digest_u32(be.external_product32(entry["log_n"], k, moduli, entry["log_basis"], glwe, ggsw_of)), {}
    if kind == "base_convert":
        count = entry["count"]
        residues = splitmix_rns(entry["seed"], moduli, count, 1)
        return digest(be.base_convert(moduli, [int(m) for m in entry["moduli_out"]], residues, count)), {}
    if kind == "base_convert32":
        count = entry["count"]
        residues = splitmix_rns(entry["seed"], moduli, count, 1).astype(np.uint32)
        return digest_u32(be.base_convert32(moduli, [int(m) for m in entry["moduli_out"]], residues, count)), {}
    if kind in ("rns_compose", "gadget_digits"):
        count = entry["count"]
        residues = splitmix_rns(entry["seed"], moduli, count, 1)
        composed, digits, shape = be.compose_and_digits(moduli, entry.get("log_basis"), residues, count)
        if kind == "rns_compose":
            return digest(composed), {}
        return digest(digits), {"decompose_length": shape[0], "drop_bits": shape[1]}
    raise KeyError(kind)


def check_reference_file(path, be):
    """Every digest of a reference_digests.json against backend `be`: list of human-readable mismatches (empty = pinned)."""
    doc = json.load(open(path))
    assert isinstance(doc.get("source"), str) and doc["source"], "reference file must name its source"
    problems = []
    for entry in doc["digests"]:
        tag = f"{entry.get('kind')}#{entry.get('case')}"
        try:
            got, extra = compute(entry, be)
        except KeyError as e:
            problems.append(f"{tag}: unknown kind or missing field {e}")
            continue
        if got != entry["output_sha256"]:
            problems.append(f"{tag}: {be.name} output {got[:16]}… != reference {entry['output_sha256'][:16]}…")
        for k, v in extra.items():
            if k in entry and str(entry[k]) != str(v):
                problems.append(f"{tag}: field {k}: {be.name} {v} != reference {entry[k]}")
    return problems, doc


def all_cases():
    return json.load(open(os.path.join(HERE, "golden", "digests.json"))) + EXTRA_CASES


# ------------------------------------------------------------------------------------------ the real file
needs_file = pytest.mark.skipif(not os.path.exists(REFERENCE_FILE),
                                reason="tests/golden/reference_digests.json absent: no cargo in the build image "
                                       "(INTEGRATION.md, 'Pinning the oracle'); parity stays unpinned until it exists")


@needs_file
def test_oracle_reproduces_the_reference_binaries_digests(orc):
    problems, doc = check_reference_file(REFERENCE_FILE, OracleBackend(orc))
    assert doc["source"].startswith("primus-fhe @"), doc["source"]
    kinds = {(e["kind"], e["case"]) for e in doc["digests"]}
    assert {(c["kind"], c["case"]) for c in all_cases()} <= kinds, "reference file does not cover every committed case"
    assert not problems, "\n".join(problems)


@needs_file
@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_binaries_digests():
    import primus_fhe_amd as pf
    problems, _ = check_reference_file(REFERENCE_FILE, HipBackend(pf))
    assert not problems, "\n".join(problems)


# ------------------------------------------------------------------------------------------ the consumer, exercised now
def handmade(orc, cases, source):
    be = OracleBackend(orc)
    out = []
    for c in cases:
        e = dict(c)
        e["output_sha256"], extra = compute(e, be)
        e.update({k: v for k, v in extra.items() if k not in e})
        out.append(e)
    return {"source": source, "generator": "tests/test_reference_goldens.py (oracle, NOT the reference)", "digests": out}


SMALL = [c for c in all_cases() if c.get("log_n", 0) <= 12 or c["kind"] in ("rns_compose", "gadget_digits", "base_convert", "base_convert32")]


def test_consumer_accepts_a_faithful_file_and_reports_a_tampered_one(orc, tmp_path):
    doc = handmade(orc, SMALL, "hand-made from the oracle (consumer self-test)")
    assert {e["kind"] for e in doc["digests"]} == {"ntt_forward", "dcrt_polymul", "external_product", "ntt32_forward",
                                                   "rns_compose", "gadget_digits", "base_convert", "external_product32",
                                                   "base_convert32"}
    good = tmp_path / "reference_digests.json"
    good.write_text(json.dumps(doc, indent=1))
    problems, _ = check_reference_file(str(good), OracleBackend(orc))
    assert problems == []
    # the committed digests of the same cases are what a faithful file must contain (same schema as digests.json)
    committed = {(d["kind"], d["case"]): d["output_sha256"] for d in json.load(open(os.path.join(HERE, "golden", "digests.json")))}
    for e in doc["digests"]:
        if (e["kind"], e["case"]) in committed:
            assert e["output_sha256"] == committed[(e["kind"], e["case"])]
    # tampering: one flipped digest, one wrong root, one unknown kind -> three reports, nothing else
    bad = json.loads(json.dumps(doc))
    bad["digests"][0]["output_sha256"] = "0" * 64
    ntt = next(e for e in bad["digests"][1:] if e["kind"] == "ntt_forward")
    ntt["root"] = "12345"
    bad["digests"].append({"kind": "fft_forward", "case": 0, "output_sha256": "0" * 64})
    badf = tmp_path / "tampered.json"
    badf.write_text(json.dumps(bad))
    problems, _ = check_reference_file(str(badf), OracleBackend(orc))
    assert len(problems) == 3 and "unknown kind" in problems[-1], problems
    # a file without a source is refused outright
    nosrc = tmp_path / "nosource.json"
    nosrc.write_text(json.dumps({"digests": []}))
    with pytest.raises(AssertionError):
        check_reference_file(str(nosrc), OracleBackend(orc))


@pytest.mark.gpu
def test_consumer_runs_the_hip_backend_on_a_handmade_file(orc, tmp_path):
    """The `-m gpu` half of the consumer on every case (full sizes): a file written from the oracle must be reproduced by
    the HIP path through the C ABI — the same comparison the real file will get."""
    import primus_fhe_amd as pf
    doc = handmade(orc, all_cases(), "hand-made from the oracle (consumer self-test)")
    f = tmp_path / "reference_digests.json"
    f.write_text(json.dumps(doc))
    problems, _ = check_reference_file(str(f), HipBackend(pf))
    assert problems == []


def test_rust_generator_and_python_consumer_list_the_same_cases():
    """integration/emit_golden/src/main.rs cannot be compiled here; what CAN be checked is that its case tables are the ones
    the consumer regenerates (shapes, moduli, batches, seeds) — a drifted table would produce a reference file whose entries
    the tests then recompute from other inputs."""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "integration", "emit_golden", "src", "main.rs")).read()
    consts = {"Q62": pyref.Q62, "Q61[0]": Q61[0], "Q61[1]": Q61[1], "Q61[2]": Q61[2]}
    assert f"const Q62: u64 = {pyref.Q62};" in src
    assert "const Q61: [u64; 3] = [%s];" % ", ".join(str(q) for q in Q61) in src
    assert "0x5EED_0000_0000_0000" in src and "0x9E37_79B9_7F4A_7C15" in src      # the SplitMix64 stream of golden_inputs.py

    def tuples(pattern):
        m = re.search(pattern + r"\s*\[(.*?)\]\s*(?:;|\.iter\(\))", src, re.S)
        assert m, pattern
        body = m.group(1)
        for k, v in consts.items():
            body = body.replace(k, str(v))
        body = re.sub(r"(\d)(u32|u64|usize)", r"\1", body)
        return [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\(([^()]*)\)", body)]

    committed = json.load(open(os.path.join(HERE, "golden", "digests.json")))
    ntt = tuples(r"let ntt_cases: \[\(u32, u64, usize\); 6\] =\s*")
    assert ntt == [(d["log_n"], int(d["q"]), d["batch"]) for d in committed if d["kind"] == "ntt_forward"]
    assert all(d["seed"] == 0x500 + d["case"] for d in committed if d["kind"] == "ntt_forward") and "0x500 + cid as u64" in src
    poly = tuples(r"for \(cid, &\(log_n, batch\)\) in ")
    assert poly == [(d["log_n"], d["batch"]) for d in committed if d["kind"] == "dcrt_polymul"]
    assert "(0x600 + cid as u64, 0x610 + cid as u64)" in src
    assert all((d["seed_a"], d["seed_b"]) == (0x600 + d["case"], 0x610 + d["case"]) for d in committed if d["kind"] == "dcrt_polymul")
    ext = tuples(r"for \(cid, &\(log_n, k, log_basis, batch\)\) in ")
    assert ext == [(d["log_n"], d["k"], d["log_basis"], d["batch"]) for d in committed if d["kind"] == "external_product"]
    assert "(0x700 + cid as u64, 0x710 + cid as u64)" in src
    assert all((d["seed_glwe"], d["seed_ggsw"]) == (0x700 + d["case"], 0x710 + d["case"])
               for d in committed if d["kind"] == "external_product")
    u32 = tuples(r"for \(cid, &\(log_n, q, batch\)\) in ")
    assert u32 == [(c["log_n"], int(c["q"]), c["batch"]) for c in EXTRA_CASES if c["kind"] == "ntt32_forward"]
    assert "0x810 + cid as u64" in src and all(c["seed"] == 0x810 + c["case"] for c in EXTRA_CASES if c["kind"] == "ntt32_forward")
    gad = tuples(r"for \(cid, &\(log_basis, count\)\) in ")
    assert gad == [(c["log_basis"], c["count"]) for c in EXTRA_CASES if c["kind"] == "gadget_digits" and c["case"] < 2]
    assert "0x300 + 0x40 + cid as u64" in src
    assert all(c["seed"] == 0x340 + c["case"] for c in EXTRA_CASES if c["kind"] in ("rns_compose", "gadget_digits"))
    # round 6: the nine-modulus base (cases 2: `cid = wid + 2`), its conversion, and the u32 external product
    assert "const W9: [u64; 9] = [%s];" % ", ".join(str(q) for q in W9) in src
    assert "const Q60: [u64; 3] = [%s];" % ", ".join(str(q) for q in Q60) in src
    assert "const Q30: [u32; 3] = [%s];" % ", ".join(str(q) for q in Q30) in src
    wide = tuples(r"for \(wid, &\(log_basis, count\)\) in ")
    assert "let cid = wid + 2;" in src
    assert wide == [(c["log_basis"], c["count"]) for c in EXTRA_CASES if c["kind"] == "gadget_digits" and c["case"] >= 2]
    conv = [c for c in EXTRA_CASES if c["kind"] == "base_convert"]
    assert len(conv) == 1 and conv[0]["count"] == wide[0][1] and conv[0]["seed"] == 0x342
    ext32 = tuples(r"for \(cid32, &\(log_n, k, log_basis, batch\)\) in ")
    assert ext32 == [(c["log_n"], c["k"], c["log_basis"], c["batch"]) for c in EXTRA_CASES if c["kind"] == "external_product32"]
    assert "(0x720 + cid as u64, 0x730 + cid as u64)" in src
    assert all((c["seed_glwe"], c["seed_ggsw"]) == (0x720 + c["case"], 0x730 + c["case"])
               for c in EXTRA_CASES if c["kind"] == "external_product32")
    # BaseConverter<u32>: the 30-bit triple into the two moduli of the reference's u32 test
    assert "const P27: [u32; 2] = [%s];" % ", ".join(str(q) for q in P27) in src
    conv32 = [c for c in EXTRA_CASES if c["kind"] == "base_convert32"]
    assert len(conv32) == 1 and "let (count, seed) = (%dusize, 0x%xu64);" % (conv32[0]["count"], conv32[0]["seed"]) in src
    # every kind the generator writes is one the consumer knows
    kinds = set(re.findall(r'\\"kind\\": \\"([a-z0-9_]+)\\"', src))
    assert kinds == {"ntt_forward", "dcrt_polymul", "external_product", "ntt32_forward", "rns_compose", "gadget_digits",
                     "base_convert", "external_product32", "base_convert32"}
