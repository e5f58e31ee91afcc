// pfhe_rns_host.cpp — host construction of RNS / gadget constants (once per parameter set).
//
// Computes what RNSBase::new (primus_rns/src/base.rs:79-117) and
// BigUintApproxSignedBasis::new (primus_decompose/src/big_integer/basis.rs:40-211) compute.
#include <algorithm>
#include <cstdlib>

#include "pfhe_rns.hpp"

namespace pfhe {

using u128 = unsigned __int128;
using Big = std::vector<u64>;  // little-endian limbs

static u64 gcd64(u64 a, u64 b) {
    while (b) {
        u64 t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static void big_trim(Big &a) {
    while (a.size() > 1 && a.back() == 0) a.pop_back();
}

static Big big_mul_u64(const Big &a, u64 v) {
    Big r(a.size() + 1, 0);
    u64 carry = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        u128 p = (u128)a[i] * v + carry;
        r[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
    r[a.size()] = carry;
    big_trim(r);
    return r;
}

static u64 big_mod_u64(const Big &a, u64 q) {
    u128 r = 0;
    for (size_t i = a.size(); i-- > 0;) r = ((r << 64) | a[i]) % q;
    return (u64)r;
}

static int big_cmp(const Big &a, const Big &b) {  // equal lengths
    for (size_t i = a.size(); i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

static void big_shl(Big &a, u32 bits) {  // in place, fixed length, overflow discarded
    const size_t n = a.size();
    const size_t words = bits / 64;
    const u32 rem = bits % 64;
    if (words) {
        for (size_t i = n; i-- > 0;) a[i] = i >= words ? a[i - words] : 0;
    }
    if (rem) {
        for (size_t i = n; i-- > 0;) a[i] = (a[i] << rem) | (i ? a[i - 1] >> (64 - rem) : 0);
    }
}

static void big_add_u64(Big &a, u64 v) {
    for (size_t i = 0; i < a.size() && v; ++i) {
        u64 s = a[i] + v;
        v = s < a[i];
        a[i] = s;
    }
}

static void big_sub(Big &a, const Big &b) {  // a -= b (a >= b)
    u64 borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        u128 d = (u128)a[i] - (i < b.size() ? b[i] : 0) - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}

static u64 inv_mod(u64 a, u64 q) {
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr != 0) {
        __int128 k = r / nr, tmp = t - k * nt;
        t = nt;
        nt = tmp;
        tmp = r - k * nr;
        r = nr;
        nr = tmp;
    }
    if (r != 1) return 0;
    if (t < 0) t += q;
    return (u64)t;
}

int build_rns(const u64 *moduli, size_t count, RnsHost &out, u32 word_bits) {
    if (count == 0) return PFHE_ERR_EMPTY_BASE;  // base.rs:47-49
    for (size_t i = 0; i < count; ++i) {
        // BarrettModulus::<T>::new requires 1 < q < 2^(T::BITS - 2) (primus_modulus/src/barrett/mod.rs:39-44)
        if (moduli[i] <= 1 || moduli[i] >= (1ull << (word_bits - 2))) {
            set_last_error(word_bits == 64 ? "RNS modulus must satisfy 1 < q < 2^62" : "RNS modulus must satisfy 1 < q < 2^30");
            return PFHE_ERR_UNREPRESENTABLE_MODULUS;
        }
    }
    for (size_t i = 0; i < count; ++i)
        for (size_t j = i + 1; j < count; ++j)
            if (gcd64(moduli[i], moduli[j]) != 1) return PFHE_ERR_COPRIME;  // base.rs:83-89
    if (count > (size_t)kMaxWideLimbs) {
        set_last_error("more than 32 RNS moduli are not supported by the device kernels");
        return PFHE_ERR_UNSUPPORTED;
    }
    Big Q{moduli[0]};
    for (size_t i = 1; i < count; ++i) Q = big_mul_u64(Q, moduli[i]);
    const size_t len = Q.size();
    out.moduli.assign(moduli, moduli + count);
    out.Q = Q;
    out.punct.assign(count * len, 0);
    out.inv_punct.assign(count, 0);
    out.inv_punct_p.assign(count, 0);
    out.ratio_lo.assign(count, 0);
    out.ratio_hi.assign(count, 0);
    for (size_t i = 0; i < count; ++i) {
        const u128 top = ((u128)1 << 64);  // floor(2^128 / q): high word, then the remainder carried down
        out.ratio_hi[i] = (u64)(top / moduli[i]);
        out.ratio_lo[i] = (u64)(((top % moduli[i]) << 64) / moduli[i]);
        Big P{1};
        for (size_t j = 0; j < count; ++j)
            if (j != i) P = big_mul_u64(P, moduli[j]);
        std::copy(P.begin(), P.end(), out.punct.begin() + i * len);
        const u64 inv = inv_mod(big_mod_u64(P, moduli[i]), moduli[i]);
        out.inv_punct[i] = inv;
        out.inv_punct_p[i] = (u64)(((u128)inv << 64) / moduli[i]);
    }
    RnsDev d{};
    d.L = (u32)count;
    d.value_len = (u32)len;
    // limbs of Q in the caller's word type (multiply_many_values, big_integer.rs:675-686: as many as Q needs)
    d.value_words = word_bits == 64 ? (u32)len : (u32)(2 * len - ((Q[len - 1] >> 32) == 0 ? 1 : 0));
    out.word_bits = word_bits;
    if (count <= (size_t)kMaxLimbs) {  // the by-value form; wider bases get their device table from upload_rns_wide
        for (size_t j = 0; j < len; ++j) d.Q[j] = Q[j];
        for (size_t i = 0; i < count; ++i) {
            d.q[i] = moduli[i];
            for (size_t j = 0; j < len; ++j) d.punct[i][j] = out.punct[i * len + j];
            d.inv_punct[i] = out.inv_punct[i];
            d.inv_punct_p[i] = out.inv_punct_p[i];
        }
    }
    u64 qmin = moduli[0], qmax = moduli[0];
    for (size_t i = 1; i < count; ++i) {
        qmin = std::min(qmin, moduli[i]);
        qmax = std::max(qmax, moduli[i]);
    }
    if ((count == 2 || count == 3) && qmax < 2 * qmin && std::getenv("PFHE_DISABLE_GARNER") == nullptr) {
        d.garner = 1;
        for (size_t i = 1; i < count; ++i)
            for (size_t j = 0; j < i; ++j) {
                const u64 inv = inv_mod(moduli[j] % moduli[i], moduli[i]);
                d.g_inv[i][j] = inv;
                d.g_inv_p[i][j] = (u64)(((u128)inv << 64) / moduli[i]);
            }
        const u128 p01 = (u128)moduli[0] * moduli[1];
        d.g_prod[0] = (u64)p01;
        d.g_prod[1] = (u64)(p01 >> 64);
    }
    out.par = RnsParams{};
    out.par.dev = d;
    return PFHE_OK;
}

int build_basis(const RnsHost &rns, u32 log_basis, size_t reverse_length, BasisHost &out) {
    const u32 len = rns.par.dev.value_len, L = rns.par.dev.L;
    if (log_basis == 0 || log_basis >= rns.word_bits) {  // basis.rs:51: 0 < log_basis < T::BITS
        set_last_error(rns.word_bits == 64 ? "log_basis must be in 1..63" : "log_basis must be in 1..31");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    const Big &Q = rns.Q;
    const u32 unused = (u32)__builtin_clzll(Q[len - 1]);
    const u32 bits = 64 * len - unused;
    size_t ell = bits / log_basis;
    u32 drop = bits - (u32)ell * log_basis;
    if (reverse_length) {  // basis.rs:63-68
        if (ell < reverse_length) {
            set_last_error("reverse_length exceeds the full decomposition length");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        ell = reverse_length;
        drop = bits - (u32)reverse_length * log_basis;
    }
    if (ell == 0) {
        set_last_error("decomposition length is zero");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    const u64 B = 1ull << log_basis, bm1 = B - 1;
    BasisDev d{};
    d.value_len = len;
    d.value_words = rns.par.dev.value_words;
    d.ell = (u32)ell;
    d.log_basis = log_basis;
    d.drop_bits = drop;
    d.basis = B;
    d.basis_minus_one = bm1;
    d.carry_mask = log_basis == 1 ? 2ull : (B | (B >> 1));  // basis.rs:81-85
    if (drop > 0) {                                          // basis.rs:72-79
        d.mode |= 1u;
        d.carry_index = (drop - 1) / 64;
        d.carry_bit_mask = 1ull << ((drop - 1) % 64);
    }
    // split value (basis.rs:87-131)
    Big t(len, 0);
    bool have = false;
    if (log_basis == 1) {
        if (drop != 0) {
            for (size_t i = 0; i <= ell; ++i) {
                big_shl(t, 1);
                t[0] |= 1;
            }
            big_shl(t, drop - 1);
            have = true;
        }
    } else {
        for (size_t i = 0; i < ell; ++i) {
            big_shl(t, log_basis);
            t[0] |= bm1 >> 1;
        }
        if (drop > 0) {
            big_shl(t, 1);
            t[0] |= 1;
            big_shl(t, drop - 1);
        } else {
            big_add_u64(t, 1);
        }
        have = true;
    }
    out.threshold.assign(len, 0);
    out.add.assign(len, 0);
    if (have && big_cmp(t, Q) < 0) {
        d.mode |= 2u;
        out.threshold = t;
        // add = (2^bits - 1) - (Q - 1)  (basis.rs:137-147)
        Big a(len, ~0ull);
        a[len - 1] >>= unused;
        Big qm1 = Q;
        Big one{1};
        big_sub(qm1, one);
        big_sub(a, qm1);
        out.add = a;
    }
    if (len <= (u32)kMaxLimbs) {  // the by-value form; wider values get their device table from upload_basis_wide
        for (u32 j = 0; j < len; ++j) {
            d.threshold[j] = out.threshold[j];
            d.add[j] = out.add[j];
        }
    }
    out.par = BasisParams{};
    out.par.dev = d;
    out.rns = rns.par;
    out.Q = Q;
    out.device = rns.device;
    out.word_bits = rns.word_bits;
    out.scalars.assign(ell * len, 0);
    out.scalars_residue.assign(ell * L, 0);
    Big s(len, 0);
    s[0] = 1;
    big_shl(s, drop);
    for (size_t j = 0; j < ell; ++j) {  // basis.rs:149-173
        std::copy(s.begin(), s.end(), out.scalars.begin() + j * len);
        for (u32 i = 0; i < L; ++i) out.scalars_residue[j * L + i] = big_mod_u64(s, rns.moduli[i]);
        big_shl(s, log_basis);
    }
    return PFHE_OK;
}

}  // namespace pfhe
