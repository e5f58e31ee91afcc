//! Shim == reference, on the committed input streams.  UNTESTED SOURCE (no Rust toolchain in the build image).
//!
//! `cargo test -p primus_ntt_hip --test emit_golden` (on a machine with a GPU, libpfhe_hip.so built and PFHE_LIB_DIR set)
//! runs every transform case of integration/emit_golden twice — through the REAL reference table and through the HIP
//! shim behind the same trait — and requires identical words; it then writes tests/golden/reference_digests.json by
//! calling the generator itself (PFHE_REPO = path of this repository), so the same command produces the pin
//! tests/test_reference_goldens.py consumes.  The stand-alone generator (`cargo run -p emit_golden`) needs no GPU.
//!
//! Reference items: `NttTable` crates/primus_ntt/src/ntt/mod.rs:16-113, `DcrtTable` crates/primus_ntt/src/dcrt/mod.rs:19-135,
//! `U64NttTable` ntt/prime64/table.rs:41, `U64DcrtTable` dcrt/prime64.rs:11, `U32NttTable` ntt/prime32/table.rs:37,
//! `prime64/tests.rs:109-122` (the reference's own "two implementations agree" test, which this mirrors).
#[path = "../../emit_golden/src/main.rs"]
#[allow(dead_code)]
mod emit;

use primus_modulus::BarrettModulus;
use primus_ntt::{DcrtTable, NttTable, U32NttTable, U64DcrtTable, U64NttTable};
use primus_ntt_hip::{HipDcrtTable, HipNttTable, HipU32NttTable};

#[test]
fn hip_tables_equal_the_reference_tables_on_the_golden_streams() {
    // U64NttTable: canonical forward / inverse words, the root, and the monomial shortcut
    for (cid, &(log_n, q, batch)) in
        [(10u32, emit::Q62, 2usize), (12, 1125899906826241, 2), (14, emit::Q61[0], 2), (16, emit::Q61[0], 1)].iter().enumerate()
    {
        let n = 1usize << log_n;
        let m = BarrettModulus::new(q);
        let (reference, hip) = (U64NttTable::new(log_n, m).unwrap(), HipNttTable::new(log_n, m).unwrap());
        assert_eq!(reference.root(), hip.root(), "minimal primitive root, case {cid}");
        let input = emit::splitmix_uniform(0x500 + cid as u64, q, n * batch);
        for poly in input.chunks_exact(n) {
            let (mut a, mut b) = (poly.to_vec(), poly.to_vec());
            reference.transform_slice(&mut a);
            hip.transform_slice(&mut b);
            assert_eq!(a, b, "forward, case {cid}");
            reference.inverse_transform_slice(&mut a);
            hip.inverse_transform_slice(&mut b);
            assert_eq!(a, b, "inverse, case {cid}");
            assert_eq!(a, poly, "round trip, case {cid}");
            // lazy outputs: representatives may differ between backends, residues may not (prime64/tests.rs:100-106)
            let (mut la, mut lb) = (poly.to_vec(), poly.to_vec());
            reference.lazy_transform_slice(&mut la);
            hip.lazy_transform_slice(&mut lb);
            assert!(la.iter().zip(&lb).all(|(x, y)| x % q == y % q && *y < 4 * q), "lazy forward, case {cid}");
        }
        let (mut ma, mut mb) = (vec![0u64; n], vec![0u64; n]);
        reference.transform_monomial(q - 2, n + 3, &mut ma);
        hip.transform_monomial(q - 2, n + 3, &mut mb);
        assert_eq!(ma, mb, "monomial, case {cid}");
    }
    // U64DcrtTable: modulus-major RNS polynomials
    for (cid, &(log_n, batch)) in [(10u32, 2usize), (16, 1)].iter().enumerate() {
        let n = 1usize << log_n;
        let moduli: Vec<BarrettModulus<u64>> = emit::Q61.iter().map(|&q| BarrettModulus::new(q)).collect();
        let (reference, hip) = (U64DcrtTable::new(log_n, &moduli).unwrap(), HipDcrtTable::new(log_n, &moduli).unwrap());
        assert_eq!(reference.crt_poly_length(), hip.crt_poly_length());
        let mut a = emit::splitmix_rns(0x600 + cid as u64, &emit::Q61, n, batch);
        let mut b = a.clone();
        reference.transform_slice(&mut a);
        hip.transform_slice(&mut b);
        assert_eq!(a, b, "DCRT forward, case {cid}");
        reference.inverse_transform_slice(&mut a);
        hip.inverse_transform_slice(&mut b);
        assert_eq!(a, b, "DCRT inverse, case {cid}");
    }
    // U32NttTable
    for (cid, &(log_n, q, batch)) in [(10u32, 132120577u32, 2usize), (16, 1073479681, 1)].iter().enumerate() {
        let n = 1usize << log_n;
        let m = BarrettModulus::new(q);
        let (reference, hip) = (U32NttTable::new(log_n, m).unwrap(), HipU32NttTable::new(log_n, m).unwrap());
        let input: Vec<u32> =
            emit::splitmix_uniform(0x810 + cid as u64, q as u64, n * batch).into_iter().map(|v| v as u32).collect();
        for poly in input.chunks_exact(n) {
            let (mut a, mut b) = (poly.to_vec(), poly.to_vec());
            reference.transform_slice(&mut a);
            hip.transform_slice(&mut b);
            assert_eq!(a, b, "u32 forward, case {cid}");
        }
    }
}

/// Writes tests/golden/reference_digests.json of the repository at $PFHE_REPO (the reference crates only: no GPU involved).
#[test]
fn emit_reference_digests() {
    let Ok(repo) = std::env::var("PFHE_REPO") else {
        eprintln!("PFHE_REPO not set: reference_digests.json not written");
        return;
    };
    let rev = std::env::var("PRIMUS_FHE_REV").unwrap_or_else(|_| "unknown revision".into());
    emit::emit_to(&format!("{repo}/tests/golden/reference_digests.json"), &rev);
}
