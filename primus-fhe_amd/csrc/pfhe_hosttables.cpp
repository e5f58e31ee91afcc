// pfhe_hosttables.cpp — host-side construction of NTT tables (runs once per (N, q)).
//
// Product code (not the oracle): computes what U64NttTable::new computes
// (primus_ntt/src/ntt/prime64/table.rs:308-405) in the layout the HIP kernels consume.
#include "pfhe_common.hpp"

namespace pfhe {

using u128 = unsigned __int128;

static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

static u64 powmod(u64 b, u64 e, u64 q) {
    u64 r = 1 % q;
    b %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod(r, b, q);
        b = mulmod(b, b, q);
    }
    return r;
}

static inline u64 shoup_quotient(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

static inline u32 bitrev(u32 x, u32 bits) {
    return bits == 0 ? 0 : (__builtin_bitreverse32(x) >> (32 - bits));
}

// Smallest element of order exactly 2^log_degree in (Z/q)^*.  The reference reaches the same
// value by a random search followed by a walk over all odd powers
// (primus_ntt/src/root.rs:60-125); the minimum does not depend on the generator found.
static int minimal_root_of_unity(u32 log_degree, u64 q, u64 &root) {
    if (q < 3 || log_degree == 0 || log_degree >= 63) return PFHE_ERR_NO_PRIMITIVE_ROOT;
    const u64 order = 1ull << log_degree;
    if ((q - 1) % order != 0) return PFHE_ERR_NO_PRIMITIVE_ROOT;
    const u64 cofactor = (q - 1) / order;
    u64 g = 0;
    for (u64 r = 2; r < 2 + 4096 && r < q; ++r) {
        u64 w = powmod(r, cofactor, q);
        if (w != 0 && powmod(w, order >> 1, q) == q - 1) {
            g = w;
            break;
        }
    }
    if (g == 0) return PFHE_ERR_NO_PRIMITIVE_ROOT;  // q is not a prime with 2^log_degree | q-1
    const u64 g2 = mulmod(g, g, q);
    const u64 g2p = shoup_quotient(g2, q);
    u64 best = g, cur = g;
    for (u64 i = 1; i < (order >> 1); ++i) {
        u64 t = g2 * cur - q * (u64)(((u128)g2p * cur) >> 64);
        cur = t >= q ? t - q : t;
        if (cur < best) best = cur;
    }
    root = best;
    return PFHE_OK;
}

int build_host_table(u32 log_n, u64 q, HostTable &out) {
    if (log_n > 22) {
        set_last_error("log_n > 22 is not supported by this build");
        return PFHE_ERR_UNSUPPORTED;
    }
    u64 psi = 0;
    // NttTable::new searches the root first (table.rs:312), then rejects q >= 2^62 (:318-323).
    if (int rc = minimal_root_of_unity(log_n + 1, q, psi)) {
        set_last_error("there is no primitive 2N-th root of unity modulo this modulus");
        return rc;
    }
    if (q >= (1ull << 62)) {
        set_last_error("modulus is too large for a u64 NTT table (max 62 bits)");
        return PFHE_ERR_MODULUS_TOO_LARGE;
    }

    const size_t n = (size_t)1 << log_n;
    out.log_n = log_n;
    out.q = q;
    out.root = psi;
    out.ordinal.resize(2 * n);
    const u64 psi_p = shoup_quotient(psi, q);
    u64 cur = 1;
    for (size_t k = 0; k < 2 * n; ++k) {
        out.ordinal[k] = cur;
        u64 t = psi * cur - q * (u64)(((u128)psi_p * cur) >> 64);
        cur = t >= q ? t - q : t;
    }
    out.inv_root = out.ordinal[2 * n - 1];

    out.fwd.assign(n, ulonglong2{0, 0});
    out.inv.assign(n, ulonglong2{0, 0});
    for (size_t k = 0; k < n; ++k) {  // roots[brv(k)] = psi^k
        u64 w = out.ordinal[k];
        out.fwd[bitrev((u32)k, log_n)] = ulonglong2{w, shoup_quotient(w, q)};
    }
    out.inv[0] = ulonglong2{1, shoup_quotient(1, q)};
    for (size_t k = 0; k + 1 < n; ++k) {  // inv_roots[brv(k)+1] = psi^(2N-1-k)
        u64 w = out.ordinal[2 * n - 1 - k];
        out.inv[bitrev((u32)k, log_n) + 1] = ulonglong2{w, shoup_quotient(w, q)};
    }
    // N | q-1, hence N * (q - (q-1)/N) = N*q - (q-1) == 1 (mod q)
    out.inv_n = q - (q - 1) / n;
    if (n == 1) out.inv_n = 1;
    out.inv_n_w = mulmod(out.inv_n, out.inv[n - 1].x, q);  // table.rs:397-399

    // floor(2^128 / q) by two-step long division
    u128 rem = 1;
    u128 c1 = rem << 64;
    out.bar_hi = (u64)(c1 / q);
    rem = c1 % q;
    u128 c0 = rem << 64;
    out.bar_lo = (u64)(c0 / q);
    return PFHE_OK;
}

}  // namespace pfhe
