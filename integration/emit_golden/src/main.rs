//! emit_golden — pins this repository's CPU oracle (and through it the HIP path) to the real primus-fhe crates.
//!
//! UNTESTED SOURCE: the build image has no `cargo` / `rustc`, so this file has never been compiled.  It is
//! written against the reference at the revision surveyed: every call below cites the item it uses.
//!
//! What it does: regenerates, bit for bit, the synthetic inputs of tests/golden_inputs.py (SplitMix64 streams seeded
//! 0x5EED_0000_0000_0000 + case id, masked to the modulus' bit length, rejection to [0, q)), runs the REAL reference
//! types on them —
//!   * `U64NttTable::transform_slice`                       crates/primus_ntt/src/ntt/prime64/table.rs:541-563
//!   * `U32NttTable::transform_slice`                       crates/primus_ntt/src/ntt/prime32/table.rs
//!   * `U64DcrtTable` + `DcrtPolynomial::mul_assign`        crates/primus_ntt/src/dcrt/prime64.rs:11-128,
//!                                                          crates/primus_poly/src/dcrt/mul.rs:176-187
//!   * `RNSBase::compose_multiple_values_to`                crates/primus_rns/src/base.rs:648-675
//!   * `BigUintApproxSignedBasis` balanced digits           crates/primus_decompose/src/big_integer/basis.rs:326-367,
//!                                                          big_integer/common.rs:275-325
//!   * `CrtGlwe::mul_dcrt_ggsw_to`                          crates/primus_lattice/src/glwe/crt.rs:200-227 (u64, and u32 over U32DcrtTable)
//!   * `BaseConverter::fast_convert_array` / `exact_convert_array`   crates/primus_rns/src/converter.rs:192-218, 274-364 (nine -> three moduli; u32: three -> two)
//! — for every case of tests/golden/digests.json (and the u32 / RNS-gadget digest cases), and writes
//! tests/golden/reference_digests.json: the same entries (same field names, SHA-256 of the little-endian output words)
//! plus `"source": "primus-fhe @ <git rev>"`.  tests/test_reference_goldens.py then requires the oracle (CPU suite) and the
//! HIP path (`-m gpu`) to reproduce that file; while it is absent those tests are skipped and DESIGN.md says
//! "parity unpinned".
//!
//! The case table below mirrors tests/golden/make_golden.py::digests(), ::u32_cases() and the digest cases added for
//! RNS composition and gadget digits (tests/test_reference_goldens.py::EXTRA_CASES) — keep the three in step.
use std::fmt::Write as _;

use primus_decompose::big_integer::BigUintApproxSignedBasis;
use primus_lattice::context::DcrtGlevContext;
use primus_lattice::ggsw::DcrtGgsw;
use primus_lattice::glwe::{CrtGlwe, DcrtGlwe};
use primus_modulus::BarrettModulus;
use primus_ntt::{DcrtTable, NttTable, U32DcrtTable, U32NttTable, U64DcrtTable, U64NttTable};
use primus_poly::DcrtPolynomial;
use primus_rns::{BaseConverter, RNSBase};

pub const SEED_BASE: u64 = 0x5EED_0000_0000_0000;
pub const Q62: u64 = 4611686018425815041;
pub const Q61: [u64; 3] = [2305843009211596801, 2305843009210023937, 2305843009208713217];
// round 6: a base of NINE moduli (the nine largest primes below 2^61 that are 1 mod 32: tests/primes.py::ntt_primes_below(9, 61, 4)),
// the three largest such primes below 2^60 as a conversion target, and the u32 tables' 30-bit triple
pub const W9: [u64; 9] = [2305843009213693921, 2305843009213693153, 2305843009213692737, 2305843009213692097, 2305843009213691041, 2305843009213690657, 2305843009213689601, 2305843009213689377, 2305843009213689089];
pub const Q60: [u64; 3] = [1152921504606845473, 1152921504606844513, 1152921504606844417];
pub const Q30: [u32; 3] = [1073479681, 1071513601, 1070727169];
pub const P27: [u32; 2] = [134215681, 134176769]; // the moduli of primus_decompose/tests/big_uint.rs:21

// ---------------------------------------------------------------------------------------------------------------
// tests/golden_inputs.py restated
// ---------------------------------------------------------------------------------------------------------------

/// word `index` (0-based) of the SplitMix64 stream `seed`: golden_inputs.splitmix_words
pub fn splitmix_word(seed: u64, index: u64) -> u64 {
    let mut z = seed.wrapping_add(0x9E37_79B9_7F4A_7C15u64.wrapping_mul(index + 1));
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}

/// first `count` accepted draws in [0, q) of stream SEED_BASE + case_id: golden_inputs.splitmix_uniform
pub fn splitmix_uniform(case_id: u64, q: u64, count: usize) -> Vec<u64> {
    let bits = 64 - q.leading_zeros();
    let mask = if bits == 64 { u64::MAX } else { (1u64 << bits) - 1 };
    let seed = SEED_BASE.wrapping_add(case_id);
    let mut out = Vec::with_capacity(count);
    let mut index = 0u64;
    while out.len() < count {
        let w = splitmix_word(seed, index) & mask;
        index += 1;
        if w < q {
            out.push(w);
        }
    }
    out
}

/// `batch` RNS polynomials, modulus-major inside each element; limb (e, r) uses stream (case_id << 20) + e * L + r:
/// golden_inputs.splitmix_rns
pub fn splitmix_rns(case_id: u64, moduli: &[u64], n: usize, batch: usize) -> Vec<u64> {
    let l = moduli.len() as u64;
    let mut out = Vec::with_capacity(batch * moduli.len() * n);
    for e in 0..batch as u64 {
        for (r, &q) in moduli.iter().enumerate() {
            out.extend(splitmix_uniform((case_id << 20) + e * l + r as u64, q, n));
        }
    }
    out
}

// ---------------------------------------------------------------------------------------------------------------
// SHA-256 of little-endian words (no external crate: the workspace has none for it)
// ---------------------------------------------------------------------------------------------------------------
const K256: [u32; 64] = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
];

pub fn sha256(bytes: &[u8]) -> String {
    let mut h: [u32; 8] =
        [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19];
    let mut msg = bytes.to_vec();
    let bit_len = (bytes.len() as u64) * 8;
    msg.push(0x80);
    while msg.len() % 64 != 56 {
        msg.push(0);
    }
    msg.extend_from_slice(&bit_len.to_be_bytes());
    for block in msg.chunks_exact(64) {
        let mut w = [0u32; 64];
        for (i, c) in block.chunks_exact(4).enumerate() {
            w[i] = u32::from_be_bytes([c[0], c[1], c[2], c[3]]);
        }
        for i in 16..64 {
            let s0 = w[i - 15].rotate_right(7) ^ w[i - 15].rotate_right(18) ^ (w[i - 15] >> 3);
            let s1 = w[i - 2].rotate_right(17) ^ w[i - 2].rotate_right(19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16].wrapping_add(s0).wrapping_add(w[i - 7]).wrapping_add(s1);
        }
        let [mut a, mut b, mut c, mut d, mut e, mut f, mut g, mut hh] = h;
        for i in 0..64 {
            let s1 = e.rotate_right(6) ^ e.rotate_right(11) ^ e.rotate_right(25);
            let ch = (e & f) ^ (!e & g);
            let t1 = hh.wrapping_add(s1).wrapping_add(ch).wrapping_add(K256[i]).wrapping_add(w[i]);
            let s0 = a.rotate_right(2) ^ a.rotate_right(13) ^ a.rotate_right(22);
            let maj = (a & b) ^ (a & c) ^ (b & c);
            let t2 = s0.wrapping_add(maj);
            hh = g;
            g = f;
            f = e;
            e = d.wrapping_add(t1);
            d = c;
            c = b;
            b = a;
            a = t1.wrapping_add(t2);
        }
        for (x, y) in h.iter_mut().zip([a, b, c, d, e, f, g, hh]) {
            *x = x.wrapping_add(y);
        }
    }
    h.iter().map(|v| format!("{v:08x}")).collect()
}

/// golden_inputs.digest: SHA-256 of the words as little-endian u64
pub fn digest_u64(words: &[u64]) -> String {
    let mut bytes = Vec::with_capacity(words.len() * 8);
    for w in words {
        bytes.extend_from_slice(&w.to_le_bytes());
    }
    sha256(&bytes)
}

/// the u32 fixtures hash their words as little-endian u32 (tests/golden/make_golden.py::u32_cases)
pub fn digest_u32(words: &[u32]) -> String {
    let mut bytes = Vec::with_capacity(words.len() * 4);
    for w in words {
        bytes.extend_from_slice(&w.to_le_bytes());
    }
    sha256(&bytes)
}

fn strs(v: &[u64]) -> String {
    let items: Vec<String> = v.iter().map(|x| format!("\"{x}\"")).collect();
    format!("[{}]", items.join(", "))
}

// ---------------------------------------------------------------------------------------------------------------
// the cases
// ---------------------------------------------------------------------------------------------------------------
fn main() {
    let out_path = std::env::args().nth(1).expect("usage: emit_golden <path/to/reference_digests.json> [git rev]");
    let rev = std::env::args().nth(2).unwrap_or_else(|| "unknown revision".to_string());
    emit_to(&out_path, &rev);
}

/// runs every case on the reference crates and writes the JSON (also called by
/// integration/primus_ntt_hip/tests/emit_golden.rs)
pub fn emit_to(out_path: &str, rev: &str) {
    let mut digests: Vec<String> = Vec::new();

    // ---- kind "ntt_forward": U64NttTable::new + transform_slice on `batch` polynomials (make_golden.py:111-123)
    let ntt_cases: [(u32, u64, usize); 6] =
        [(10, Q62, 2), (12, 1125899906826241, 2), (14, Q61[0], 2), (16, Q61[0], 1), (16, Q61[1], 1), (16, Q61[2], 1)];
    for (cid, &(log_n, q, batch)) in ntt_cases.iter().enumerate() {
        let n = 1usize << log_n;
        let seed = 0x500 + cid as u64;
        let table = U64NttTable::new(log_n, BarrettModulus::new(q)).expect("NTT table");
        let mut x = splitmix_uniform(seed, q, n * batch);
        let input_sha = digest_u64(&x);
        for poly in x.chunks_exact_mut(n) {
            table.transform_slice(poly);
        }
        digests.push(format!(
            "{{\"kind\": \"ntt_forward\", \"case\": {cid}, \"log_n\": {log_n}, \"q\": \"{q}\", \"batch\": {batch}, \"seed\": {seed}, \
             \"root\": \"{}\", \"input_sha256\": \"{input_sha}\", \"output_sha256\": \"{}\"}}",
            table.root(),
            digest_u64(&x)
        ));
    }

    // ---- kind "dcrt_polymul": NTT both operands, DcrtPolynomial::mul_assign, inverse NTT (make_golden.py:125-147)
    for (cid, &(log_n, batch)) in [(10u32, 2usize), (16, 1)].iter().enumerate() {
        let n = 1usize << log_n;
        let moduli: Vec<BarrettModulus<u64>> = Q61.iter().map(|&q| BarrettModulus::new(q)).collect();
        let table = U64DcrtTable::new(log_n, &moduli).expect("DCRT table");
        let w = table.crt_poly_length();
        let (seed_a, seed_b) = (0x600 + cid as u64, 0x610 + cid as u64);
        let mut a = splitmix_rns(seed_a, &Q61, n, batch);
        let mut b = splitmix_rns(seed_b, &Q61, n, batch);
        for (pa, pb) in a.chunks_exact_mut(w).zip(b.chunks_exact_mut(w)) {
            table.transform_slice(pa);
            table.transform_slice(pb);
            DcrtPolynomial(&mut pa[..]).mul_assign(&DcrtPolynomial(&pb[..]), n, &moduli);
            table.inverse_transform_slice(pa);
        }
        digests.push(format!(
            "{{\"kind\": \"dcrt_polymul\", \"case\": {cid}, \"log_n\": {log_n}, \"moduli\": {}, \"batch\": {batch}, \
             \"seed_a\": {seed_a}, \"seed_b\": {seed_b}, \"output_sha256\": \"{}\"}}",
            strs(&Q61),
            digest_u64(&a)
        ));
    }

    // ---- kind "external_product": CrtGlwe::mul_dcrt_ggsw_to with one shared NTT-domain GGSW (make_golden.py:150-162)
    for (cid, &(log_n, k, log_basis, batch)) in [(10u32, 1usize, 30u32, 2usize), (16, 1, 30, 1)].iter().enumerate() {
        let n = 1usize << log_n;
        let moduli: Vec<BarrettModulus<u64>> = Q61.iter().map(|&q| BarrettModulus::new(q)).collect();
        let table = U64DcrtTable::new(log_n, &moduli).expect("DCRT table");
        let base = RNSBase::<u64, BarrettModulus<u64>>::new(&moduli).expect("RNS base");
        let basis = BigUintApproxSignedBasis::<u64>::new(base.moduli_product(), log_basis, None, &base);
        let ell = basis.decompose_length();
        let crt_len = table.crt_poly_length();
        let glwe_len = (k + 1) * crt_len;
        let (seed_glwe, seed_ggsw) = (0x700 + cid as u64, 0x710 + cid as u64);
        let glwe = splitmix_rns(seed_glwe, &Q61, n, batch * (k + 1));
        let ggsw = splitmix_rns(seed_ggsw, &Q61, n, (k + 1) * ell * (k + 1));
        let mut context =
            DcrtGlevContext::<u64>::new(n, crt_len, n * base.big_uint_value_len(), base.moduli_count());
        let mut result = vec![0u64; batch * glwe_len];
        for (ct, out) in glwe.chunks_exact(glwe_len).zip(result.chunks_exact_mut(glwe_len)) {
            CrtGlwe(ct).mul_dcrt_ggsw_to(
                &DcrtGgsw(&ggsw[..]),
                &mut DcrtGlwe(&mut out[..]),
                &basis,
                &table,
                &base,
                &mut context,
            );
        }
        digests.push(format!(
            "{{\"kind\": \"external_product\", \"case\": {cid}, \"log_n\": {log_n}, \"k\": {k}, \"moduli\": {}, \
             \"log_basis\": {log_basis}, \"batch\": {batch}, \"seed_glwe\": {seed_glwe}, \"seed_ggsw\": {seed_ggsw}, \
             \"output_sha256\": \"{}\"}}",
            strs(&Q61),
            digest_u64(&result)
        ));
    }

    // ---- kind "ntt32_forward": U32NttTable (tests/golden/u32_ntt.json "digests", make_golden.py:181-190)
    for (cid, &(log_n, q, batch)) in [(10u32, 132120577u32, 2usize), (16, 1073479681, 1)].iter().enumerate() {
        let n = 1usize << log_n;
        let seed = 0x810 + cid as u64;
        let table = U32NttTable::new(log_n, BarrettModulus::new(q)).expect("u32 NTT table");
        let mut x: Vec<u32> = splitmix_uniform(seed, q as u64, n * batch).into_iter().map(|v| v as u32).collect();
        for poly in x.chunks_exact_mut(n) {
            table.transform_slice(poly);
        }
        digests.push(format!(
            "{{\"kind\": \"ntt32_forward\", \"case\": {cid}, \"log_n\": {log_n}, \"q\": \"{q}\", \"batch\": {batch}, \"seed\": {seed}, \
             \"output_sha256\": \"{}\"}}",
            digest_u32(&x)
        ));
    }

    // ---- kinds "rns_compose" and "gadget_digits": RNSBase::compose_multiple_values_to, then the balanced digits of
    //      every level (init_value_carry_slice_inplace + unsigned_decompose_slice_to), least significant level first,
    //      each digit as a u64 in [0, B) (tests/test_reference_goldens.py::EXTRA_CASES)
    for (cid, &(log_basis, count)) in [(30u32, 4096usize), (13, 1000)].iter().enumerate() {
        let moduli: Vec<BarrettModulus<u64>> = Q61.iter().map(|&q| BarrettModulus::new(q)).collect();
        let base = RNSBase::<u64, BarrettModulus<u64>>::new(&moduli).expect("RNS base");
        let value_len = base.big_uint_value_len();
        let seed = 0x300 + 0x40 + cid as u64;
        let residues = splitmix_rns(seed, &Q61, count, 1);
        let mut values = vec![0u64; count * value_len];
        let mut scratch = vec![0u64; base.moduli_count()];
        base.compose_multiple_values_to(&residues, &mut values, count, &mut scratch);
        digests.push(format!(
            "{{\"kind\": \"rns_compose\", \"case\": {cid}, \"moduli\": {}, \"count\": {count}, \"seed\": {seed}, \
             \"output_sha256\": \"{}\"}}",
            strs(&Q61),
            digest_u64(&values)
        ));
        let basis = BigUintApproxSignedBasis::<u64>::new(base.moduli_product(), log_basis, None, &base);
        let mut carries = vec![false; count];
        basis.init_value_carry_slice_inplace(&mut values, &mut carries, value_len);
        let mut all_digits: Vec<u64> = Vec::with_capacity(count * basis.decompose_length());
        let mut level = vec![0u64; count];
        for decomposer in basis.decomposer_iter() {
            decomposer.unsigned_decompose_slice_to(&values, &mut level, &mut carries, value_len);
            all_digits.extend_from_slice(&level);
        }
        digests.push(format!(
            "{{\"kind\": \"gadget_digits\", \"case\": {cid}, \"moduli\": {}, \"log_basis\": {log_basis}, \"count\": {count}, \
             \"seed\": {seed}, \"decompose_length\": {}, \"drop_bits\": {}, \"output_sha256\": \"{}\"}}",
            strs(&Q61),
            basis.decompose_length(),
            basis.drop_bits(),
            digest_u64(&all_digits)
        ));
    }

    // ---- round 6, wide base (nine moduli): kinds "rns_compose" / "gadget_digits" case 2 on W9, and kind "base_convert":
    //      BaseConverter::fast_convert_array W9 -> Q60 (converter.rs:192-218) and exact_convert_array W9 -> Q60[0]
    //      (converter.rs:274-364) of the same residues
    for (wid, &(log_basis, count)) in [(30u32, 2048usize)].iter().enumerate() {
        let cid = wid + 2;
        let moduli: Vec<BarrettModulus<u64>> = W9.iter().map(|&q| BarrettModulus::new(q)).collect();
        let base = RNSBase::<u64, BarrettModulus<u64>>::new(&moduli).expect("RNS base of nine moduli");
        let value_len = base.big_uint_value_len();
        let seed = 0x300 + 0x40 + cid as u64;
        let residues = splitmix_rns(seed, &W9, count, 1);
        let mut values = vec![0u64; count * value_len];
        let mut scratch = vec![0u64; base.moduli_count()];
        base.compose_multiple_values_to(&residues, &mut values, count, &mut scratch);
        digests.push(format!(
            "{{\"kind\": \"rns_compose\", \"case\": {cid}, \"moduli\": {}, \"count\": {count}, \"seed\": {seed}, \
             \"output_sha256\": \"{}\"}}",
            strs(&W9),
            digest_u64(&values)
        ));
        let basis = BigUintApproxSignedBasis::<u64>::new(base.moduli_product(), log_basis, None, &base);
        let mut carries = vec![false; count];
        basis.init_value_carry_slice_inplace(&mut values, &mut carries, value_len);
        let mut all_digits: Vec<u64> = Vec::with_capacity(count * basis.decompose_length());
        let mut level = vec![0u64; count];
        for decomposer in basis.decomposer_iter() {
            decomposer.unsigned_decompose_slice_to(&values, &mut level, &mut carries, value_len);
            all_digits.extend_from_slice(&level);
        }
        digests.push(format!(
            "{{\"kind\": \"gadget_digits\", \"case\": {cid}, \"moduli\": {}, \"log_basis\": {log_basis}, \"count\": {count}, \
             \"seed\": {seed}, \"decompose_length\": {}, \"drop_bits\": {}, \"output_sha256\": \"{}\"}}",
            strs(&W9),
            basis.decompose_length(),
            basis.drop_bits(),
            digest_u64(&all_digits)
        ));
        let out_moduli: Vec<BarrettModulus<u64>> = Q60.iter().map(|&q| BarrettModulus::new(q)).collect();
        let out_base = RNSBase::<u64, BarrettModulus<u64>>::new(&out_moduli).expect("output base");
        let conv = BaseConverter::new(&base, &out_base);
        let mut fast = vec![0u64; count * Q60.len()];
        let mut conv_scratch = vec![0u64; count * W9.len()];
        conv.fast_convert_array(&residues, &mut fast, count, &mut conv_scratch);
        let one_base = RNSBase::<u64, BarrettModulus<u64>>::new(&out_moduli[..1]).expect("one-modulus base");
        let exact_conv = BaseConverter::new(&base, &one_base);
        let mut exact = vec![0u64; count];
        exact_conv.exact_convert_array(&residues, &mut exact, count);
        fast.extend_from_slice(&exact);
        digests.push(format!(
            "{{\"kind\": \"base_convert\", \"case\": 0, \"moduli\": {}, \"moduli_out\": {}, \"count\": {count}, \"seed\": {seed}, \
             \"output_sha256\": \"{}\"}}",
            strs(&W9),
            strs(&Q60),
            digest_u64(&fast)
        ));
    }

    // ---- round 6, kind "external_product32": CrtGlwe::<u32>::mul_dcrt_ggsw_to over U32DcrtTable (dcrt/prime32.rs:11) with
    //      RNSBase<u32, BarrettModulus<u32>> and BigUintApproxSignedBasis<u32>; inputs are the u64 streams narrowed to u32
    for (cid32, &(log_n, k, log_basis, batch)) in [(10u32, 1usize, 15u32, 2usize)].iter().enumerate() {
        let cid = cid32;
        let n = 1usize << log_n;
        let q30_wide: Vec<u64> = Q30.iter().map(|&q| q as u64).collect();
        let moduli: Vec<BarrettModulus<u32>> = Q30.iter().map(|&q| BarrettModulus::new(q)).collect();
        let table = U32DcrtTable::new(log_n, &moduli).expect("u32 DCRT table");
        let base = RNSBase::<u32, BarrettModulus<u32>>::new(&moduli).expect("u32 RNS base");
        let basis = BigUintApproxSignedBasis::<u32>::new(base.moduli_product(), log_basis, None, &base);
        let ell = basis.decompose_length();
        let crt_len = table.crt_poly_length();
        let glwe_len = (k + 1) * crt_len;
        let (seed_glwe, seed_ggsw) = (0x720 + cid as u64, 0x730 + cid as u64);
        let glwe: Vec<u32> = splitmix_rns(seed_glwe, &q30_wide, n, batch * (k + 1)).into_iter().map(|v| v as u32).collect();
        let ggsw: Vec<u32> = splitmix_rns(seed_ggsw, &q30_wide, n, (k + 1) * ell * (k + 1)).into_iter().map(|v| v as u32).collect();
        let mut context =
            DcrtGlevContext::<u32>::new(n, crt_len, n * base.big_uint_value_len(), base.moduli_count());
        let mut result = vec![0u32; batch * glwe_len];
        for (ct, out) in glwe.chunks_exact(glwe_len).zip(result.chunks_exact_mut(glwe_len)) {
            CrtGlwe(ct).mul_dcrt_ggsw_to(
                &DcrtGgsw(&ggsw[..]),
                &mut DcrtGlwe(&mut out[..]),
                &basis,
                &table,
                &base,
                &mut context,
            );
        }
        digests.push(format!(
            "{{\"kind\": \"external_product32\", \"case\": {cid}, \"log_n\": {log_n}, \"k\": {k}, \"moduli\": {}, \
             \"log_basis\": {log_basis}, \"batch\": {batch}, \"seed_glwe\": {seed_glwe}, \"seed_ggsw\": {seed_ggsw}, \
             \"output_sha256\": \"{}\"}}",
            strs(&q30_wide),
            digest_u32(&result)
        ));
    }

    // ---- round 6, kind "base_convert32": BaseConverter::<u32, BarrettModulus<u32>> (converter.rs:21, generic over T) from
    //      the 30-bit triple into the two moduli of the reference's own u32 test (primus_decompose/tests/big_uint.rs:21):
    //      fast_convert_array, then exact_convert_array into the first of them, of the same residues
    {
        let (count, seed) = (2048usize, 0x350u64);
        let q30_wide: Vec<u64> = Q30.iter().map(|&q| q as u64).collect();
        let in_moduli: Vec<BarrettModulus<u32>> = Q30.iter().map(|&q| BarrettModulus::new(q)).collect();
        let out_moduli: Vec<BarrettModulus<u32>> = P27.iter().map(|&q| BarrettModulus::new(q)).collect();
        let base = RNSBase::<u32, BarrettModulus<u32>>::new(&in_moduli).expect("u32 input base");
        let out_base = RNSBase::<u32, BarrettModulus<u32>>::new(&out_moduli).expect("u32 output base");
        let residues: Vec<u32> = splitmix_rns(seed, &q30_wide, count, 1).into_iter().map(|v| v as u32).collect();
        let conv = BaseConverter::new(&base, &out_base);
        let mut fast = vec![0u32; count * P27.len()];
        let mut conv_scratch = vec![0u32; count * Q30.len()];
        conv.fast_convert_array(&residues, &mut fast, count, &mut conv_scratch);
        let one_base = RNSBase::<u32, BarrettModulus<u32>>::new(&out_moduli[..1]).expect("one-modulus base");
        let exact_conv = BaseConverter::new(&base, &one_base);
        let mut exact = vec![0u32; count];
        exact_conv.exact_convert_array(&residues, &mut exact, count);
        fast.extend_from_slice(&exact);
        let p27_wide: Vec<u64> = P27.iter().map(|&q| q as u64).collect();
        digests.push(format!(
            "{{\"kind\": \"base_convert32\", \"case\": 0, \"moduli\": {}, \"moduli_out\": {}, \"count\": {count}, \"seed\": {seed}, \
             \"output_sha256\": \"{}\"}}",
            strs(&q30_wide),
            strs(&p27_wide),
            digest_u32(&fast)
        ));
    }

    let mut json = String::new();
    writeln!(json, "{{").unwrap();
    writeln!(json, " \"source\": \"primus-fhe @ {rev}\",").unwrap();
    writeln!(json, " \"generator\": \"integration/emit_golden (cargo run --release -p emit_golden)\",").unwrap();
    writeln!(json, " \"digests\": [").unwrap();
    for (i, d) in digests.iter().enumerate() {
        writeln!(json, "  {d}{}", if i + 1 == digests.len() { "" } else { "," }).unwrap();
    }
    writeln!(json, " ]").unwrap();
    writeln!(json, "}}").unwrap();
    std::fs::write(out_path, json).expect("write reference_digests.json");
    eprintln!("wrote {} digests to {out_path}", digests.len());
}
