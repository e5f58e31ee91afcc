/*
 * pfhe_oracle.c — CPU restatement (plain C, gcc) of the primus-fhe hot path.
 *
 * TEST INFRASTRUCTURE ONLY — see pfhe_oracle.h for the scope statement and the
 * "parity unpinned" note.  Citations are file:line under /root/reference/crates/.
 */
#include "pfhe_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ========================================================================== */
/* scalar arithmetic — primus_ntt/src/ntt/prime64/scalar/arithmetic.rs          */
/* ========================================================================== */

/* arithmetic.rs:3-5  x.min(x.wrapping_sub(q)) */
uint64_t orc_reduce_once(uint64_t x, uint64_t q) {
    uint64_t y = x - q;
    return x < y ? x : y;
}

/* arithmetic.rs:9-12 */
uint64_t orc_reduce_twice(uint64_t x, uint64_t q, uint64_t two_q) {
    return orc_reduce_once(orc_reduce_once(x, two_q), q);
}

/* primus_factor/src/shoup_factor/mod.rs:35-44,65-67: floor(w * 2^64 / q)
 * (DivWide::div_wide(lo=0, hi=w, q), primus_integer/src/integer_traits/division.rs:49-62) */
uint64_t orc_shoup_quotient(uint64_t w, uint64_t q) {
    return (uint64_t)((((u128)w) << 64) / q);
}

/* primus_factor/src/mul_factor/mod.rs:18 — floor(w * 2^shift / q) */
static uint64_t multiply_factor_quotient(uint64_t w, uint32_t shift, uint64_t q) {
    return (uint64_t)((((u128)w) << shift) / q);
}
uint64_t orc_multiply_factor_quotient(uint64_t w, uint32_t shift, uint64_t q) { return multiply_factor_quotient(w, shift, q); }

/* arithmetic.rs:32-35  Barrett-64 lazy multiply, result in [0, 2q) */
uint64_t orc_mul_mod_lazy(uint64_t y, uint64_t w, uint64_t w_precon, uint64_t q) {
    uint64_t qhat = (uint64_t)(((u128)y * (u128)w_precon) >> 64);
    return w * y - q * qhat;
}

/* arithmetic.rs:23-28  Barrett-32 lazy multiply for q < 2^30 */
uint64_t orc_mul_mod_lazy32(uint64_t y, uint64_t w, uint64_t w_precon32, uint64_t q) {
    uint32_t qhat = (uint32_t)((y * w_precon32) >> 32);
    return (uint64_t)(uint32_t)((uint32_t)w * (uint32_t)y - (uint32_t)q * qhat);
}

/* shoup_factor/mod.rs:124-143: lazy then min(t, t - q) */
uint64_t orc_shoup_mul(uint64_t w, uint64_t w_precon, uint64_t b, uint64_t q) {
    uint64_t t = orc_mul_mod_lazy(b, w, w_precon, q);
    return orc_reduce_once(t, q);
}

/* arithmetic.rs:43-59 */
static inline void fwd_butterfly(uint64_t *x, uint64_t *y, uint64_t w, uint64_t wp, uint64_t q,
                                 uint64_t two_q, uint32_t bit_shift) {
    uint64_t tx = orc_reduce_once(*x, two_q);
    uint64_t t = bit_shift == 32 ? orc_mul_mod_lazy32(*y, w, wp, q) : orc_mul_mod_lazy(*y, w, wp, q);
    *x = tx + t;
    *y = tx + two_q - t;
}

/* arithmetic.rs:63-79 */
static inline void inv_butterfly(uint64_t *x, uint64_t *y, uint64_t w, uint64_t wp, uint64_t q,
                                 uint64_t two_q, uint32_t bit_shift) {
    uint64_t tx = *x + *y;
    uint64_t y_red = *x + two_q - *y;
    *x = orc_reduce_once(tx, two_q);
    *y = bit_shift == 32 ? orc_mul_mod_lazy32(y_red, w, wp, q) : orc_mul_mod_lazy(y_red, w, wp, q);
}

/* ========================================================================== */
/* BarrettModulus<u64> — primus_modulus/src/barrett/{mod.rs,ops.rs}             */
/* ========================================================================== */

/* mod.rs:39-59: ratio = floor(2^128 / value); requires 1 < value < 2^62 */
int orc_barrett_new(uint64_t value, orc_barrett *out) {
    if (value <= 1) return ORC_ERR_BAD_ARG;
    if (__builtin_clzll(value) <= 1) return ORC_ERR_MODULUS_TOO_LARGE;
    /* long division of [0,0,1] (little endian limbs) by value */
    u128 rem = 1; /* top limb */
    uint64_t qd[2];
    for (int i = 1; i >= 0; --i) {
        u128 cur = rem << 64; /* next limb is 0 */
        qd[i] = (uint64_t)(cur / value);
        rem = cur % value;
    }
    out->value = value;
    out->ratio[0] = qd[0];
    out->ratio[1] = qd[1];
    return ORC_OK;
}

/* mod.rs:99-132 — keeps exactly the partial-product structure of the reference */
uint64_t orc_barrett_lazy_reduce_wide(const orc_barrett *m, uint64_t lo, uint64_t hi) {
    uint64_t ah = (uint64_t)(((u128)lo * m->ratio[0]) >> 64);       /* widening_mul_hw */
    u128 b = (u128)lo * m->ratio[1] + ah;                           /* carrying_mul */
    u128 c = (u128)hi * m->ratio[0];                                /* widening_mul */
    uint64_t d = hi * m->ratio[1];                                  /* wrapping_mul */
    uint64_t b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    uint64_t c0 = (uint64_t)c, c1 = (uint64_t)(c >> 64);
    uint64_t carry = (uint64_t)(b0 + c0 < b0);                      /* overflowing_add(..).1 */
    uint64_t bch = b1 + c1 + carry;                                 /* carrying_add(..).0 */
    uint64_t qq = d + bch;
    return lo - qq * m->value;
}

uint64_t orc_barrett_reduce_wide(const orc_barrett *m, uint64_t lo, uint64_t hi) {
    return orc_reduce_once(orc_barrett_lazy_reduce_wide(m, lo, hi), m->value);
}

/* ops.rs:13-33: single-word lazy reduce then reduce_once */
uint64_t orc_barrett_reduce(const orc_barrett *m, uint64_t v) {
    uint64_t tmp = (uint64_t)(((u128)v * m->ratio[0]) >> 64);
    uint64_t qq = (uint64_t)(((u128)v * m->ratio[1] + tmp) >> 64);
    return orc_reduce_once(v - qq * m->value, m->value);
}

/* ops.rs:276-283 */
uint64_t orc_barrett_mul(const orc_barrett *m, uint64_t a, uint64_t b) {
    u128 p = (u128)a * b;
    return orc_barrett_reduce_wide(m, (uint64_t)p, (uint64_t)(p >> 64));
}

/* ops.rs:308-315: reduce(a.carrying_mul(b, c)) */
uint64_t orc_barrett_mul_add(const orc_barrett *m, uint64_t a, uint64_t b, uint64_t c) {
    u128 p = (u128)a * b + c;
    return orc_barrett_reduce_wide(m, (uint64_t)p, (uint64_t)(p >> 64));
}

/* primus_modulus/src/common/compact/primitive.rs:10-13 */
uint64_t orc_reduce_add(uint64_t q, uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t t = s - q;
    return s < t ? s : t;
}

/* primitive.rs:36-39 */
uint64_t orc_reduce_sub(uint64_t q, uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    uint64_t t = d + q;
    return d < t ? d : t;
}

uint64_t orc_pow_mod(uint64_t base, uint64_t exp, uint64_t q) {
    u128 r = 1 % q, b = base % q;
    while (exp) {
        if (exp & 1) r = (r * b) % q;
        b = (b * b) % q;
        exp >>= 1;
    }
    return (uint64_t)r;
}

/* primus_gcd/src/lib.rs:124 (Xgcd::gcdinv) — any correct inverse gives the same canonical value */
uint64_t orc_inv_mod(uint64_t a, uint64_t q) {
    __int128 t = 0, newt = 1;
    __int128 r = q, newr = a % q;
    while (newr != 0) {
        __int128 quo = r / newr;
        __int128 tmp = t - quo * newt; t = newt; newt = tmp;
        tmp = r - quo * newr; r = newr; newr = tmp;
    }
    if (r != 1) return 0; /* not invertible */
    if (t < 0) t += q;
    return (uint64_t)t;
}

/* compact/slice.rs:106-115 bound to BarrettModulus at barrett/slice.rs:247-294 */
void orc_reduce_mul_slice_assign(uint64_t q, uint64_t *a, const uint64_t *b, size_t n) {
    orc_barrett m;
    if (orc_barrett_new(q, &m)) return;
    for (size_t i = 0; i < n; ++i) a[i] = orc_barrett_mul(&m, a[i], b[i]);
}

/* compact/slice.rs:210-221: acc = a*b + acc mod q */
void orc_reduce_add_mul_slice_assign(uint64_t q, uint64_t *acc, const uint64_t *a,
                                     const uint64_t *b, size_t n) {
    orc_barrett m;
    if (orc_barrett_new(q, &m)) return;
    for (size_t i = 0; i < n; ++i) acc[i] = orc_barrett_mul_add(&m, a[i], b[i], acc[i]);
}

/* ========================================================================== */
/* primitive root — primus_ntt/src/root.rs                                      */
/* ========================================================================== */

/* root.rs:60-125.  The reference samples r at random (root.rs:83-99); the result of
 * try_minimal_primitive_root is the minimum over ALL primitive 2^log_degree-th roots
 * (root.rs:107-124) and therefore independent of which generator was found, so a
 * deterministic scan r = 2,3,... restates it exactly. */
int orc_minimal_primitive_root(uint32_t log_degree, uint64_t q, uint64_t *root) {
    if (log_degree == 0 || log_degree >= 64 || q < 3) return ORC_ERR_NO_PRIMITIVE_ROOT;
    uint64_t qm1 = q - 1;
    uint64_t degree = 1ull << log_degree;
    uint64_t quotient = qm1 >> log_degree;
    if (qm1 != quotient * degree) return ORC_ERR_NO_PRIMITIVE_ROOT; /* root.rs:76-81 */

    uint64_t w = 0;
    int found = 0;
    for (uint64_t r = 2; r < q && r < 2 + 4096; ++r) {
        w = orc_pow_mod(r, quotient, q);
        /* is_primitive_root (root.rs:41-58): w != 0 and w^(degree/2) == q-1 */
        if (w != 0 && orc_pow_mod(w, degree >> 1, q) == qm1) { found = 1; break; }
    }
    if (!found) return ORC_ERR_NO_PRIMITIVE_ROOT;

    /* root.rs:107-122: walk w * (w^2)^k, k < degree, keep the minimum */
    uint64_t gsq = (uint64_t)(((u128)w * w) % q);
    uint64_t gsq_p = orc_shoup_quotient(gsq, q);
    uint64_t cur = w, best = w;
    for (uint64_t i = 0; i < degree; ++i) {
        if (cur < best) best = cur;
        cur = orc_shoup_mul(gsq, gsq_p, cur, q);
    }
    *root = best;
    return ORC_OK;
}

static size_t reverse_lsbs(size_t x, uint32_t bits) { /* primus_ntt/src/reverse.rs */
    size_t r = 0;
    for (uint32_t i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* ========================================================================== */
/* U64NttTable — primus_ntt/src/ntt/prime64/table.rs                            */
/* ========================================================================== */

struct orc_u64_ntt {
    size_t n;
    uint32_t log_n;
    uint64_t q, two_q;
    int low_q;
    uint64_t root, inv_root;
    uint64_t inv_n, inv_n_precon32, inv_n_precon64;
    uint64_t inv_n_w, inv_n_w_precon32, inv_n_w_precon64;
    uint64_t *roots, *roots_precon32, *roots_precon64, *roots_precon52;
    uint64_t *inv_roots, *inv_roots_precon32, *inv_roots_precon64, *inv_roots_precon52;
    uint64_t *ordinal_roots; /* 2n */
    size_t *rev;             /* reverse_lsbs, n */
};

/* table.rs:308-405 */
int orc_u64_ntt_new(uint32_t log_n, uint64_t q, orc_u64_ntt **out) {
    uint64_t root;
    int rc = orc_minimal_primitive_root(log_n + 1, q, &root); /* :312 */
    if (rc) return rc;
    if (q >= (1ull << 62)) return ORC_ERR_MODULUS_TOO_LARGE; /* :318-323 */

    orc_u64_ntt *t = (orc_u64_ntt *)calloc(1, sizeof(*t));
    size_t n = (size_t)1 << log_n;
    t->n = n; t->log_n = log_n; t->q = q; t->two_q = q << 1;
    t->low_q = q < (1ull << 30);
    t->root = root;

    /* :329-338 ordinal roots [1, w, ..., w^(2n-1)] via ShoupFactor */
    uint64_t root_p = orc_shoup_quotient(root, q);
    t->ordinal_roots = (uint64_t *)malloc(2 * n * sizeof(uint64_t));
    t->ordinal_roots[0] = 1;
    t->ordinal_roots[1] = root;
    uint64_t power = root;
    for (size_t i = 2; i < 2 * n; ++i) {
        power = orc_shoup_mul(root, root_p, power, q);
        t->ordinal_roots[i] = power;
    }
    t->inv_root = t->ordinal_roots[2 * n - 1]; /* :340 */

    t->rev = (size_t *)malloc(n * sizeof(size_t)); /* :344 */
    for (size_t i = 0; i < n; ++i) t->rev[i] = reverse_lsbs(i, log_n);

    /* :347-351 forward roots, bit-reversed */
    t->roots = (uint64_t *)calloc(n, sizeof(uint64_t));
    t->roots[0] = 1;
    for (size_t k = 0; k < n; ++k) t->roots[t->rev[k]] = t->ordinal_roots[k];

    /* :354-358 inverse roots: ordinal[n+1..].rev() zipped with reverse_lsbs, stored at i+1 */
    t->inv_roots = (uint64_t *)calloc(n, sizeof(uint64_t));
    t->inv_roots[0] = 1;
    for (size_t k = 0; k + 1 < n; ++k) t->inv_roots[t->rev[k] + 1] = t->ordinal_roots[2 * n - 1 - k];

    /* :364-385 preconditioners */
    t->roots_precon64 = (uint64_t *)malloc(n * sizeof(uint64_t));
    t->inv_roots_precon64 = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) {
        t->roots_precon64[i] = orc_shoup_quotient(t->roots[i], q);
        t->inv_roots_precon64[i] = orc_shoup_quotient(t->inv_roots[i], q);
    }
    if (q < (1ull << 50)) { /* Barrett-52 preconditioners of the IFMA rung (table.rs:98-110, precompute.rs:37-45) */
        t->roots_precon52 = (uint64_t *)malloc(n * sizeof(uint64_t));
        t->inv_roots_precon52 = (uint64_t *)malloc(n * sizeof(uint64_t));
        for (size_t i = 0; i < n; ++i) {
            t->roots_precon52[i] = multiply_factor_quotient(t->roots[i], 52, q);
            t->inv_roots_precon52[i] = multiply_factor_quotient(t->inv_roots[i], 52, q);
        }
    }
    if (t->low_q) {
        t->roots_precon32 = (uint64_t *)malloc(n * sizeof(uint64_t));
        t->inv_roots_precon32 = (uint64_t *)malloc(n * sizeof(uint64_t));
        for (size_t i = 0; i < n; ++i) {
            t->roots_precon32[i] = multiply_factor_quotient(t->roots[i], 32, q);
            t->inv_roots_precon32[i] = multiply_factor_quotient(t->inv_roots[i], 32, q);
        }
    }

    /* :388-405 */
    t->inv_n = orc_inv_mod((uint64_t)n, q);
    t->inv_n_precon64 = orc_shoup_quotient(t->inv_n, q);
    t->inv_n_precon32 = t->low_q ? (t->inv_n << 32) / q : 0;
    uint64_t last_w = t->inv_roots[n - 1];
    t->inv_n_w = orc_reduce_once(orc_mul_mod_lazy(last_w, t->inv_n, t->inv_n_precon64, q), q);
    t->inv_n_w_precon64 = orc_shoup_quotient(t->inv_n_w, q);
    t->inv_n_w_precon32 = t->low_q ? (t->inv_n_w << 32) / q : 0;

    *out = t;
    return ORC_OK;
}

void orc_u64_ntt_free(orc_u64_ntt *t) {
    if (!t) return;
    free(t->roots); free(t->roots_precon32); free(t->roots_precon64); free(t->roots_precon52);
    free(t->inv_roots); free(t->inv_roots_precon32); free(t->inv_roots_precon64); free(t->inv_roots_precon52);
    free(t->ordinal_roots); free(t->rev);
    free(t);
}

size_t orc_u64_ntt_n(const orc_u64_ntt *t) { return t->n; }
uint64_t orc_u64_ntt_modulus(const orc_u64_ntt *t) { return t->q; }
uint64_t orc_u64_ntt_root(const orc_u64_ntt *t) { return t->root; }
uint64_t orc_u64_ntt_inv_root(const orc_u64_ntt *t) { return t->inv_root; }
uint64_t orc_u64_ntt_inv_n(const orc_u64_ntt *t) { return t->inv_n; }
uint64_t orc_u64_ntt_inv_n_w(const orc_u64_ntt *t) { return t->inv_n_w; }
const uint64_t *orc_u64_ntt_roots(const orc_u64_ntt *t) { return t->roots; }
const uint64_t *orc_u64_ntt_roots_precon64(const orc_u64_ntt *t) { return t->roots_precon64; }
const uint64_t *orc_u64_ntt_inv_roots(const orc_u64_ntt *t) { return t->inv_roots; }
const uint64_t *orc_u64_ntt_inv_roots_precon64(const orc_u64_ntt *t) { return t->inv_roots_precon64; }
const uint64_t *orc_u64_ntt_roots_precon52(const orc_u64_ntt *t) { return t->roots_precon52; }
const uint64_t *orc_u64_ntt_roots_precon32(const orc_u64_ntt *t) { return t->roots_precon32; }
const uint64_t *orc_u64_ntt_inv_roots_precon32(const orc_u64_ntt *t) { return t->inv_roots_precon32; }
const uint64_t *orc_u64_ntt_inv_roots_precon52(const orc_u64_ntt *t) { return t->inv_roots_precon52; }
const uint64_t *orc_u64_ntt_ordinal_roots(const orc_u64_ntt *t) { return t->ordinal_roots; }

static uint32_t pick_shift(const orc_u64_ntt *t, uint32_t bit_shift) {
    if (bit_shift == 32 || bit_shift == 64) return bit_shift;
    return t->low_q ? 32 : 64; /* table.rs:226-230 */
}

/* scalar/transform.rs:13-141.  The reference special-cases t = 8,4,2,1 only to unroll; the
 * arithmetic of every arm is the generic arm's (lines 126-136), which is what is restated,
 * plus the fused final reduction of the t == 1 arm (lines 104-116). */
void orc_u64_ntt_scalar_forward(const orc_u64_ntt *tb, uint64_t *values, uint32_t bit_shift,
                                uint32_t output_mod_factor) {
    bit_shift = pick_shift(tb, bit_shift);
    const size_t n = tb->n;
    const uint64_t q = tb->q, two_q = tb->two_q;
    const uint64_t *roots = tb->roots;
    const uint64_t *precon = bit_shift == 32 ? tb->roots_precon32 : tb->roots_precon64;
    size_t ri = 1; /* skip roots[0] */
    size_t t = n >> 1;
    for (size_t m = 1; m < n; m <<= 1, t >>= 1) {
        for (size_t c = 0; c < m; ++c) {
            uint64_t w = roots[ri], wp = precon[ri];
            ++ri;
            uint64_t *xs = values + 2 * t * c, *ys = xs + t;
            for (size_t j = 0; j < t; ++j) {
                fwd_butterfly(&xs[j], &ys[j], w, wp, q, two_q, bit_shift);
                if (t == 1 && output_mod_factor == 1) {
                    xs[j] = orc_reduce_twice(xs[j], q, two_q);
                    ys[j] = orc_reduce_twice(ys[j], q, two_q);
                }
            }
        }
    }
}

/* scalar/transform.rs:151-319 */
void orc_u64_ntt_scalar_inverse(const orc_u64_ntt *tb, uint64_t *values, uint32_t bit_shift,
                                uint32_t output_mod_factor) {
    bit_shift = pick_shift(tb, bit_shift);
    const size_t n = tb->n;
    const uint64_t q = tb->q, two_q = tb->two_q;
    const uint64_t *inv_roots = tb->inv_roots;
    const uint64_t *precon = bit_shift == 32 ? tb->inv_roots_precon32 : tb->inv_roots_precon64;
    const uint64_t inv_n = tb->inv_n, inv_n_w = tb->inv_n_w;
    const uint64_t inv_n_p = bit_shift == 32 ? tb->inv_n_precon32 : tb->inv_n_precon64;
    const uint64_t inv_n_w_p = bit_shift == 32 ? tb->inv_n_w_precon32 : tb->inv_n_w_precon64;

    size_t ri = 1, t = 1;
    for (size_t m = n >> 1; m > 1; m >>= 1, t <<= 1) { /* :189-281 */
        for (size_t c = 0; c < m; ++c) {
            uint64_t w = inv_roots[ri], wp = precon[ri];
            ++ri;
            uint64_t *xs = values + 2 * t * c, *ys = xs + t;
            for (size_t j = 0; j < t; ++j) inv_butterfly(&xs[j], &ys[j], w, wp, q, two_q, bit_shift);
        }
    }
    /* :283-318 final stage fused with inv_n / inv_n_w */
    uint64_t *xs = values, *ys = values + n / 2;
    for (size_t j = 0; j < n / 2; ++j) {
        uint64_t tx = orc_reduce_once(xs[j] + ys[j], two_q);
        uint64_t ty = xs[j] + two_q - ys[j];
        uint64_t rx, ry;
        if (bit_shift == 32) {
            rx = orc_mul_mod_lazy32(tx, inv_n, inv_n_p, q);
            ry = orc_mul_mod_lazy32(ty, inv_n_w, inv_n_w_p, q);
        } else {
            rx = orc_mul_mod_lazy(tx, inv_n, inv_n_p, q);
            ry = orc_mul_mod_lazy(ty, inv_n_w, inv_n_w_p, q);
        }
        if (output_mod_factor == 1) { rx = orc_reduce_once(rx, q); ry = orc_reduce_once(ry, q); }
        xs[j] = rx; ys[j] = ry;
    }
}

/* table.rs:541-563 */
void orc_u64_ntt_lazy_transform_slice(const orc_u64_ntt *t, uint64_t *p) { orc_u64_ntt_scalar_forward(t, p, 0, 4); }
/* table.rs:166-302: U64NttTable dispatches to its vector backend when the host has one.  The restatement
 * keeps the scalar path as the default (it is what the parity tests pin) and switches the canonical
 * forward transform to the AVX-512 DQ restatement (pfhe_oracle_avx512.c) only on request — used by
 * bench.py's cpu_baseline so that the CPU figure is what the reference would achieve on that host. */
static int g_vector_backend = 0;
void orc_set_vector_backend(int on) { g_vector_backend = on && orc_avx512_available(); }
int orc_get_vector_backend(void) { return g_vector_backend; }
void orc_u64_ntt_transform_slice(const orc_u64_ntt *t, uint64_t *p) {
    if (g_vector_backend && t->n >= 16 && orc_u64_ntt_forward_avx512(t, p, 0) == ORC_OK) return;
    orc_u64_ntt_scalar_forward(t, p, 0, 1);
}
void orc_u64_ntt_lazy_inverse_transform_slice(const orc_u64_ntt *t, uint64_t *v) { orc_u64_ntt_scalar_inverse(t, v, 0, 2); }
void orc_u64_ntt_inverse_transform_slice(const orc_u64_ntt *t, uint64_t *v) {
    if (g_vector_backend && t->n >= 16 && orc_u64_ntt_inverse_avx512(t, v, 0) == ORC_OK) return;
    orc_u64_ntt_scalar_inverse(t, v, 0, 1);
}

/* table.rs:565-609 */
void orc_u64_ntt_transform_monomial(const orc_u64_ntt *t, uint64_t coeff, size_t degree,
                                    uint64_t *values) {
    const size_t n = t->n;
    if (coeff == 0) { memset(values, 0, n * sizeof(uint64_t)); return; }
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = coeff; return; }
    const size_t mask = (2 * n) - 1; /* usize::MAX >> (BITS - log_n - 1) */
    if (coeff == 1) {
        for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
    } else if (coeff == t->q - 1) {
        for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[(((2 * t->rev[i] + 1) * degree) & mask) ^ n];
    } else {
        uint64_t cp = orc_shoup_quotient(coeff, t->q);
        for (size_t i = 0; i < n; ++i) {
            uint64_t w = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
            values[i] = orc_shoup_mul(coeff, cp, w, t->q);
        }
    }
}

/* table.rs:611-630 */
void orc_u64_ntt_transform_coeff_one_monomial(const orc_u64_ntt *t, size_t degree, uint64_t *values) {
    const size_t n = t->n;
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = 1; return; }
    const size_t mask = (2 * n) - 1;
    for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
}

/* table.rs:632-651 */
void orc_u64_ntt_transform_coeff_minus_one_monomial(const orc_u64_ntt *t, size_t degree, uint64_t *values) {
    const size_t n = t->n;
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = t->q - 1; return; }
    const size_t mask = (2 * n) - 1;
    for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[(((2 * t->rev[i] + 1) * degree) & mask) ^ n];
}

/* ========================================================================== */
/* U32NttTable — primus_ntt/src/ntt/prime32/{table.rs, scalar/arithmetic.rs, scalar/transform.rs} */
/* ========================================================================== */

/* arithmetic.rs:3-6 */
static inline uint32_t reduce_once32(uint32_t x, uint32_t q) { uint32_t d = x - q; return x < d ? x : d; }
/* arithmetic.rs:10-13 */
static inline uint32_t reduce_twice32(uint32_t x, uint32_t q, uint32_t two_q) { return reduce_once32(reduce_once32(x, two_q), q); }
/* arithmetic.rs:16-20 */
uint32_t orc_u32_mul_mod_lazy(uint32_t y, uint32_t w, uint32_t w_precon, uint32_t q) {
    uint32_t qhat = (uint32_t)(((uint64_t)y * (uint64_t)w_precon) >> 32);
    return w * y - q * qhat;
}
/* arithmetic.rs:23-36 */
static inline void fwd_butterfly32(uint32_t *x, uint32_t *y, uint32_t w, uint32_t wp, uint32_t q, uint32_t two_q) {
    uint32_t tx = reduce_once32(*x, two_q);
    uint32_t ty = orc_u32_mul_mod_lazy(*y, w, wp, q);
    *x = tx + ty;
    *y = tx + two_q - ty;
}
/* arithmetic.rs:39-51 */
static inline void inv_butterfly32(uint32_t *x, uint32_t *y, uint32_t w, uint32_t wp, uint32_t q, uint32_t two_q) {
    uint32_t tx = *x + *y;
    uint32_t ty = *x + two_q - *y;
    *x = reduce_once32(tx, two_q);
    *y = orc_u32_mul_mod_lazy(ty, w, wp, q);
}
/* ShoupFactor::<u32>::quotient_for: floor(w * 2^32 / q) (shoup_factor/mod.rs:39,65-67) */
static inline uint32_t shoup_quotient32(uint32_t w, uint32_t q) { return (uint32_t)(((uint64_t)w << 32) / q); }

struct orc_u32_ntt {
    size_t n;
    uint32_t log_n, q, two_q, root, inv_root;
    uint32_t inv_n, inv_n_precon, inv_n_w, inv_n_w_precon;
    uint32_t *roots, *roots_precon, *inv_roots, *inv_roots_precon;
    uint32_t *ordinal_roots; /* 2n */
    size_t *rev;
};

/* table.rs:184-333 */
int orc_u32_ntt_new(uint32_t log_n, uint32_t q, orc_u32_ntt **out) {
    uint64_t root64;
    int rc = orc_minimal_primitive_root(log_n + 1, q, &root64); /* :189 */
    if (rc) return rc;
    if (q >= (1u << 30)) return ORC_ERR_MODULUS_TOO_LARGE; /* :195-200 */
    const uint32_t root = (uint32_t)root64;

    orc_u32_ntt *t = (orc_u32_ntt *)calloc(1, sizeof(*t));
    const size_t n = (size_t)1 << log_n;
    t->n = n; t->log_n = log_n; t->q = q; t->two_q = q << 1; t->root = root;
    t->ordinal_roots = (uint32_t *)malloc(2 * n * sizeof(uint32_t));
    t->roots = (uint32_t *)calloc(n, sizeof(uint32_t));
    t->roots_precon = (uint32_t *)calloc(n, sizeof(uint32_t));
    t->inv_roots = (uint32_t *)calloc(n, sizeof(uint32_t));
    t->inv_roots_precon = (uint32_t *)calloc(n, sizeof(uint32_t));
    t->rev = (size_t *)malloc(n * sizeof(size_t));

    /* :205-214 ordinal powers by ShoupFactor::factor_mul_modulo */
    const uint32_t root_p = shoup_quotient32(root, q);
    t->ordinal_roots[0] = 1;
    if (2 * n > 1) t->ordinal_roots[1] = root;
    uint32_t power = root;
    for (size_t i = 2; i < 2 * n; ++i) {
        power = reduce_once32(orc_u32_mul_mod_lazy(power, root, root_p, q), q);
        t->ordinal_roots[i] = power;
    }
    t->inv_root = t->ordinal_roots[2 * n - 1]; /* :216 */
    for (size_t i = 0; i < n; ++i) t->rev[i] = reverse_lsbs(i, log_n); /* :220 */
    t->roots[0] = 1;
    for (size_t i = 0; i < n; ++i) t->roots[t->rev[i]] = t->ordinal_roots[i]; /* :223-227 */
    t->inv_roots[0] = 1;
    for (size_t i = 0; i + 1 < n; ++i) t->inv_roots[t->rev[i] + 1] = t->ordinal_roots[2 * n - 1 - i]; /* :230-234 */
    for (size_t i = 0; i < n; ++i) { /* :237-250 */
        t->roots_precon[i] = shoup_quotient32(t->roots[i], q);
        t->inv_roots_precon[i] = shoup_quotient32(t->inv_roots[i], q);
    }
    t->inv_n = (uint32_t)orc_inv_mod((uint64_t)(n % q), q); /* :253 (mod_inv via xgcd) */
    t->inv_n_precon = shoup_quotient32(t->inv_n, q);
    const uint32_t last_w = t->inv_roots[n - 1]; /* :257-259 */
    t->inv_n_w = reduce_once32(orc_u32_mul_mod_lazy(last_w, t->inv_n, t->inv_n_precon, q), q);
    t->inv_n_w_precon = (uint32_t)(((uint64_t)t->inv_n_w << 32) / q);
    *out = t;
    return ORC_OK;
}

void orc_u32_ntt_free(orc_u32_ntt *t) {
    if (!t) return;
    free(t->roots); free(t->roots_precon); free(t->inv_roots); free(t->inv_roots_precon);
    free(t->ordinal_roots); free(t->rev); free(t);
}

size_t orc_u32_ntt_n(const orc_u32_ntt *t) { return t->n; }
uint32_t orc_u32_ntt_modulus(const orc_u32_ntt *t) { return t->q; }
uint32_t orc_u32_ntt_root(const orc_u32_ntt *t) { return t->root; }
uint32_t orc_u32_ntt_inv_root(const orc_u32_ntt *t) { return t->inv_root; }
uint32_t orc_u32_ntt_inv_n(const orc_u32_ntt *t) { return t->inv_n; }
uint32_t orc_u32_ntt_inv_n_w(const orc_u32_ntt *t) { return t->inv_n_w; }
const uint32_t *orc_u32_ntt_roots(const orc_u32_ntt *t) { return t->roots; }
const uint32_t *orc_u32_ntt_inv_roots(const orc_u32_ntt *t) { return t->inv_roots; }

/* scalar/transform.rs:13-140; the t = 8/4/2/1 arms are unrolled forms of the generic arm */
void orc_u32_ntt_scalar_forward(const orc_u32_ntt *tb, uint32_t *values, uint32_t output_mod_factor) {
    const size_t n = tb->n;
    const uint32_t q = tb->q, two_q = tb->two_q;
    size_t ri = 1, t = n >> 1, m = 1;
    while (m < n) {
        for (size_t c = 0; c < n; c += 2 * t) {
            const uint32_t w = tb->roots[ri], wp = tb->roots_precon[ri];
            ++ri;
            for (size_t j = 0; j < t; ++j) fwd_butterfly32(&values[c + j], &values[c + j + t], w, wp, q, two_q);
            if (t == 1 && output_mod_factor == 1) { /* :104-114 */
                values[c] = reduce_twice32(values[c], q, two_q);
                values[c + 1] = reduce_twice32(values[c + 1], q, two_q);
            }
        }
        t >>= 1;
        m <<= 1;
    }
}

/* scalar/transform.rs:152-272 */
void orc_u32_ntt_scalar_inverse(const orc_u32_ntt *tb, uint32_t *values, uint32_t output_mod_factor) {
    const size_t n = tb->n;
    const uint32_t q = tb->q, two_q = tb->two_q;
    size_t ri = 1, t = 1, m = n >> 1;
    while (m > 1) {
        for (size_t c = 0; c < n; c += 2 * t) {
            const uint32_t w = tb->inv_roots[ri], wp = tb->inv_roots_precon[ri];
            ++ri;
            for (size_t j = 0; j < t; ++j) inv_butterfly32(&values[c + j], &values[c + j + t], w, wp, q, two_q);
        }
        t <<= 1;
        m >>= 1;
    }
    const size_t h = n / 2; /* :253-271 */
    for (size_t i = 0; i < h; ++i) {
        uint32_t *x = &values[i], *y = &values[i + h];
        uint32_t tx = reduce_once32(*x + *y, two_q);
        uint32_t ty = *x + two_q - *y;
        uint32_t rx = orc_u32_mul_mod_lazy(tx, tb->inv_n, tb->inv_n_precon, q);
        uint32_t ry = orc_u32_mul_mod_lazy(ty, tb->inv_n_w, tb->inv_n_w_precon, q);
        if (output_mod_factor == 1) { rx = reduce_once32(rx, q); ry = reduce_once32(ry, q); }
        *x = rx; *y = ry;
    }
}

/* table.rs:356-374 */
void orc_u32_ntt_lazy_transform_slice(const orc_u32_ntt *t, uint32_t *p) { orc_u32_ntt_scalar_forward(t, p, 4); }
void orc_u32_ntt_transform_slice(const orc_u32_ntt *t, uint32_t *p) { orc_u32_ntt_scalar_forward(t, p, 1); }
void orc_u32_ntt_lazy_inverse_transform_slice(const orc_u32_ntt *t, uint32_t *v) { orc_u32_ntt_scalar_inverse(t, v, 2); }
void orc_u32_ntt_inverse_transform_slice(const orc_u32_ntt *t, uint32_t *v) { orc_u32_ntt_scalar_inverse(t, v, 1); }

/* table.rs:376-426 */
void orc_u32_ntt_transform_monomial(const orc_u32_ntt *t, uint32_t coeff, size_t degree, uint32_t *values) {
    const size_t n = t->n;
    if (coeff == 0) { memset(values, 0, n * sizeof(uint32_t)); return; }
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = coeff; return; }
    const size_t mask = (2 * n) - 1;
    if (coeff == 1) {
        for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
    } else if (coeff == t->q - 1) {
        for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[(((2 * t->rev[i] + 1) * degree) & mask) ^ n];
    } else {
        const uint32_t cp = shoup_quotient32(coeff, t->q);
        for (size_t i = 0; i < n; ++i) {
            uint32_t w = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
            values[i] = reduce_once32(orc_u32_mul_mod_lazy(w, coeff, cp, t->q), t->q);
        }
    }
}

/* table.rs:428-447 */
void orc_u32_ntt_transform_coeff_one_monomial(const orc_u32_ntt *t, size_t degree, uint32_t *values) {
    const size_t n = t->n;
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = 1; return; }
    const size_t mask = (2 * n) - 1;
    for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[((2 * t->rev[i] + 1) * degree) & mask];
}

/* table.rs:449-470 */
void orc_u32_ntt_transform_coeff_minus_one_monomial(const orc_u32_ntt *t, size_t degree, uint32_t *values) {
    const size_t n = t->n;
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = t->q - 1; return; }
    const size_t mask = (2 * n) - 1;
    for (size_t i = 0; i < n; ++i) values[i] = t->ordinal_roots[(((2 * t->rev[i] + 1) * degree) & mask) ^ n];
}

/* Pointwise product / multiply-accumulate of NTT-domain u32 polynomials: BarrettModulus<u32>
 * reduce_mul / reduce_mul_add return the canonical residue (barrett/ops.rs), i.e. a*b mod q. */
void orc_u32_reduce_mul_slice_assign(uint32_t q, uint32_t *a, const uint32_t *b, size_t n) {
    for (size_t i = 0; i < n; ++i) a[i] = (uint32_t)(((uint64_t)a[i] * b[i]) % q);
}
void orc_u32_reduce_add_mul_slice_assign(uint32_t q, uint32_t *acc, const uint32_t *a, const uint32_t *b, size_t n) {
    for (size_t i = 0; i < n; ++i) acc[i] = (uint32_t)(((uint64_t)a[i] * b[i] + acc[i]) % q);
}

/* ========================================================================== */
/* UintNttTable<u64> — primus_ntt/src/ntt/primitive.rs (the reference's own oracle) */
/* ========================================================================== */

typedef struct { uint64_t value, quotient; } shoup_t; /* ShoupFactor AoS */

struct orc_uint_ntt {
    size_t n; uint32_t log_n; uint64_t q, root, inv_root;
    shoup_t inv_n;
    shoup_t *root_powers, *inv_root_powers, *ordinal; /* n, n, 2n */
    size_t *rev;
};

static shoup_t shoup_new(uint64_t v, uint64_t q) { shoup_t s = {v, orc_shoup_quotient(v, q)}; return s; }
static uint64_t shoup_lazy(shoup_t s, uint64_t b, uint64_t q) { return orc_mul_mod_lazy(b, s.value, s.quotient, q); }
static uint64_t shoup_full(shoup_t s, uint64_t b, uint64_t q) { return orc_reduce_once(shoup_lazy(s, b, q), q); }

/* primitive.rs:116-184 */
int orc_uint_ntt_new(uint32_t log_n, uint64_t q, orc_uint_ntt **out) {
    uint64_t root;
    int rc = orc_minimal_primitive_root(log_n + 1, q, &root);
    if (rc) return rc;
    size_t n = (size_t)1 << log_n;
    if ((uint64_t)n >= q) return ORC_ERR_DEGREE_TOO_LARGE; /* :166-168 */
    orc_uint_ntt *t = (orc_uint_ntt *)calloc(1, sizeof(*t));
    t->n = n; t->log_n = log_n; t->q = q; t->root = root;
    shoup_t root_factor = shoup_new(root, q);
    t->ordinal = (shoup_t *)malloc(2 * n * sizeof(shoup_t));
    t->ordinal[0] = shoup_new(1, q);
    t->ordinal[1] = root_factor;
    uint64_t power = root;
    for (size_t i = 2; i < 2 * n; ++i) {
        power = shoup_full(root_factor, power, q);
        t->ordinal[i] = shoup_new(power, q);
    }
    t->inv_root = t->ordinal[2 * n - 1].value;
    t->rev = (size_t *)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) t->rev[i] = reverse_lsbs(i, log_n);
    t->root_powers = (shoup_t *)calloc(n, sizeof(shoup_t));
    t->root_powers[0] = t->ordinal[0];
    for (size_t k = 0; k < n; ++k) t->root_powers[t->rev[k]] = t->ordinal[k];
    t->inv_root_powers = (shoup_t *)calloc(n, sizeof(shoup_t));
    t->inv_root_powers[0] = t->ordinal[0];
    for (size_t k = 0; k + 1 < n; ++k) t->inv_root_powers[t->rev[k] + 1] = t->ordinal[2 * n - 1 - k];
    t->inv_n = shoup_new(orc_inv_mod((uint64_t)n, q), q);
    *out = t;
    return ORC_OK;
}

void orc_uint_ntt_free(orc_uint_ntt *t) {
    if (!t) return;
    free(t->root_powers); free(t->inv_root_powers); free(t->ordinal); free(t->rev); free(t);
}

/* primitive.rs:209-233 */
void orc_uint_ntt_lazy_transform_slice(const orc_uint_ntt *t, uint64_t *poly) {
    const uint64_t q = t->q, two_q = q << 1;
    size_t ri = 1;
    for (int lg = (int)t->log_n - 1; lg >= 0; --lg) {
        size_t gap = (size_t)1 << lg;
        for (size_t base = 0; base < t->n; base += gap << 1) {
            shoup_t root = t->root_powers[ri++];
            for (size_t j = 0; j < gap; ++j) {
                uint64_t u = orc_reduce_once(poly[base + j], two_q);
                uint64_t v = shoup_lazy(root, poly[base + gap + j], q);
                poly[base + j] = u + v;
                poly[base + gap + j] = u + two_q - v;
            }
        }
    }
}

/* primitive.rs:235-246 */
void orc_uint_ntt_transform_slice(const orc_uint_ntt *t, uint64_t *poly) {
    orc_uint_ntt_lazy_transform_slice(t, poly);
    for (size_t i = 0; i < t->n; ++i) poly[i] = orc_reduce_once(orc_reduce_once(poly[i], t->q << 1), t->q);
}

/* primitive.rs:248-289 */
void orc_uint_ntt_lazy_inverse_transform_slice(const orc_uint_ntt *t, uint64_t *values) {
    const uint64_t q = t->q, two_q = q << 1;
    size_t ri = 1;
    for (uint32_t lg = 0; lg + 1 < t->log_n; ++lg) {
        size_t gap = (size_t)1 << lg;
        for (size_t base = 0; base < t->n; base += gap << 1) {
            shoup_t root = t->inv_root_powers[ri++];
            for (size_t j = 0; j < gap; ++j) {
                uint64_t u = values[base + j], v = values[base + gap + j];
                values[base + j] = orc_reduce_add(two_q, u, v);
                values[base + gap + j] = shoup_lazy(root, u + two_q - v, q);
            }
        }
    }
    size_t gap = (size_t)1 << (t->log_n - 1);
    shoup_t scalar = t->inv_n;
    uint64_t scaled_r = shoup_full(t->inv_root_powers[ri], scalar.value, q);
    shoup_t scaled = shoup_new(scaled_r, q);
    for (size_t j = 0; j < gap; ++j) {
        uint64_t u = values[j], v = values[gap + j];
        values[j] = shoup_full(scalar, u + v, q);
        values[gap + j] = shoup_full(scaled, u + two_q - v, q);
    }
}

/* primitive.rs:291-298 */
void orc_uint_ntt_inverse_transform_slice(const orc_uint_ntt *t, uint64_t *values) {
    orc_uint_ntt_lazy_inverse_transform_slice(t, values);
    for (size_t i = 0; i < t->n; ++i) values[i] = orc_reduce_once(values[i], t->q);
}

/* primitive.rs:300-350 */
void orc_uint_ntt_transform_monomial(const orc_uint_ntt *t, uint64_t coeff, size_t degree, uint64_t *values) {
    const size_t n = t->n;
    if (coeff == 0) { memset(values, 0, n * sizeof(uint64_t)); return; }
    if (degree == 0) { for (size_t i = 0; i < n; ++i) values[i] = coeff; return; }
    const size_t mask = 2 * n - 1;
    for (size_t i = 0; i < n; ++i) {
        size_t index = ((2 * t->rev[i] + 1) * degree) & mask;
        if (coeff == 1) values[i] = t->ordinal[index].value;
        else if (coeff == t->q - 1) values[i] = t->ordinal[index ^ n].value;
        else values[i] = shoup_full(t->ordinal[index], coeff, t->q);
    }
}

/* ========================================================================== */
/* U64DcrtTable — primus_ntt/src/dcrt/prime64.rs                                */
/* ========================================================================== */

struct orc_dcrt {
    size_t count, n;
    orc_u64_ntt **tables;
};

/* dcrt/prime64.rs:24-43 */
int orc_dcrt_new(uint32_t log_n, const uint64_t *moduli, size_t count, orc_dcrt **out) {
    orc_dcrt *d = (orc_dcrt *)calloc(1, sizeof(*d));
    d->count = count; d->n = (size_t)1 << log_n;
    d->tables = (orc_u64_ntt **)calloc(count ? count : 1, sizeof(*d->tables));
    for (size_t i = 0; i < count; ++i) {
        int rc = orc_u64_ntt_new(log_n, moduli[i], &d->tables[i]);
        if (rc) { orc_dcrt_free(d); return rc; }
    }
    *out = d;
    return ORC_OK;
}

void orc_dcrt_free(orc_dcrt *d) {
    if (!d) return;
    for (size_t i = 0; i < d->count; ++i) orc_u64_ntt_free(d->tables[i]);
    free(d->tables); free(d);
}

size_t orc_dcrt_poly_length(const orc_dcrt *t) { return t->n; }
size_t orc_dcrt_moduli_count(const orc_dcrt *t) { return t->count; }
const orc_u64_ntt *orc_dcrt_table(const orc_dcrt *t, size_t i) { return t->tables[i]; }

/* dcrt/prime64.rs:106-111: modulus-major chunks of N */
void orc_dcrt_transform_slice(const orc_dcrt *t, uint64_t *poly) {
    for (size_t i = 0; i < t->count; ++i) orc_u64_ntt_transform_slice(t->tables[i], poly + i * t->n);
}

/* dcrt/prime64.rs:122-127 */
void orc_dcrt_inverse_transform_slice(const orc_dcrt *t, uint64_t *poly) {
    for (size_t i = 0; i < t->count; ++i) orc_u64_ntt_inverse_transform_slice(t->tables[i], poly + i * t->n);
}

/* primus_poly/src/dcrt/mul.rs:176-187 */
void orc_dcrt_poly_mul_assign(const orc_dcrt *t, uint64_t *a, const uint64_t *b) {
    for (size_t i = 0; i < t->count; ++i)
        orc_reduce_mul_slice_assign(t->tables[i]->q, a + i * t->n, b + i * t->n, t->n);
}

/* primus_poly/src/dcrt/mod.rs:105-123 */
void orc_dcrt_poly_add_mul_assign(const orc_dcrt *t, uint64_t *acc, const uint64_t *a, const uint64_t *b) {
    for (size_t i = 0; i < t->count; ++i)
        orc_reduce_add_mul_slice_assign(t->tables[i]->q, acc + i * t->n, a + i * t->n, b + i * t->n, t->n);
}

/* primus_poly/src/dcrt/mul.rs:15-30 (slice_butterfly) applied per limb as in
 * DcrtPolynomial::butterfly_mul_factor_to (:196-222): a' = a + s, b = (a - s) * w, all canonical.
 * `w` holds ShoupFactor pairs (value, quotient), L*N of them. */
void orc_dcrt_poly_butterfly_mul_factor_to(const orc_dcrt *t, uint64_t *a, const uint64_t *s, const uint64_t *w_pairs,
                                           uint64_t *b) {
    for (size_t i = 0; i < t->count; ++i) {
        const uint64_t q = t->tables[i]->q;
        for (size_t j = 0; j < t->n; ++j) {
            const size_t e = i * t->n + j;
            const uint64_t a_orig = a[e];
            a[e] = orc_reduce_add(q, a_orig, s[e]);
            const uint64_t diff = orc_reduce_sub(q, a_orig, s[e]);
            b[e] = orc_shoup_mul(w_pairs[2 * e], w_pairs[2 * e + 1], diff, q);
        }
    }
}

/* DcrtPolynomial::butterfly_mul_to (primus_poly/src/dcrt/mod.rs:125-160): same with a plain
 * multiplicand and a Barrett product. */
void orc_dcrt_poly_butterfly_mul_to(const orc_dcrt *t, uint64_t *a, const uint64_t *s, const uint64_t *w, uint64_t *b) {
    for (size_t i = 0; i < t->count; ++i) {
        orc_barrett m;
        if (orc_barrett_new(t->tables[i]->q, &m)) return;
        for (size_t j = 0; j < t->n; ++j) {
            const size_t e = i * t->n + j;
            const uint64_t a_orig = a[e];
            a[e] = orc_reduce_add(m.value, a_orig, s[e]);
            b[e] = orc_barrett_mul(&m, orc_reduce_sub(m.value, a_orig, s[e]), w[e]);
        }
    }
}

/* ---------------- CrtPolynomial / DcrtPolynomial element-wise family ------------------------------
 * One RNS polynomial (L limbs of n words, modulus-major) per call; a CrtGlwe is k+1 of them and its
 * add_element_wise* / mul_scalar_* / mul_factor_to / mul_monic_monomial_assign loop the same per-limb
 * slice operations (primus_lattice/src/macros/mod.rs:367-531, glwe/crt.rs:59-175). */

/* primus_modulus/src/common/uint/primitive.rs:19-25 */
uint64_t orc_reduce_neg(uint64_t q, uint64_t v) { return v == 0 ? 0 : q - v; }

/* crt/add.rs:51-70 -> reduce_add_slice_to */
void orc_crt_poly_add_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, const uint64_t *b,
                         uint64_t *out) {
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < n; ++j) out[i * n + j] = orc_reduce_add(moduli[i], a[i * n + j], b[i * n + j]);
}

/* crt/sub.rs:47-66 -> reduce_sub_slice_to (sub_rev_assign, :69-82, is the same with out = b) */
void orc_crt_poly_sub_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, const uint64_t *b,
                         uint64_t *out) {
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < n; ++j) out[i * n + j] = orc_reduce_sub(moduli[i], a[i * n + j], b[i * n + j]);
}

/* crt/neg.rs:42-53 -> reduce_neg_slice_to */
void orc_crt_poly_neg_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, uint64_t *out) {
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < n; ++j) out[i * n + j] = orc_reduce_neg(moduli[i], a[i * n + j]);
}

/* crt/mul.rs:138-158 -> reduce_mul_scalar_slice_to (compact/slice.rs:145-153): BarrettModulus::reduce_mul */
int orc_crt_poly_mul_scalar_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a,
                               const uint64_t *scalars, uint64_t *out) {
    for (size_t i = 0; i < L; ++i) {
        orc_barrett m;
        if (orc_barrett_new(moduli[i], &m)) return 1;
        for (size_t j = 0; j < n; ++j) out[i * n + j] = orc_barrett_mul(&m, a[i * n + j], scalars[i]);
    }
    return 0;
}

/* crt/mul.rs:57-77 -> reduce_add_mul_scalar_slice_assign (compact/slice.rs:277-286):
 * acc = reduce_mul_add(a, scalar, acc) */
int orc_crt_poly_add_mul_scalar_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *acc,
                                       const uint64_t *rhs, const uint64_t *scalars) {
    for (size_t i = 0; i < L; ++i) {
        orc_barrett m;
        if (orc_barrett_new(moduli[i], &m)) return 1;
        for (size_t j = 0; j < n; ++j)
            acc[i * n + j] = orc_barrett_mul_add(&m, rhs[i * n + j], scalars[i], acc[i * n + j]);
    }
    return 0;
}

/* crt/mul.rs:161-180 -> factor_mul_slice_to (primus_factor/src/common/slice.rs:49-58): factor_mul_modulo.
 * `factors` = L (value, quotient) pairs. */
void orc_crt_poly_mul_factor_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a,
                                const uint64_t *factors, uint64_t *out) {
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < n; ++j)
            out[i * n + j] = orc_shoup_mul(factors[2 * i], factors[2 * i + 1], a[i * n + j], moduli[i]);
}

/* crt/mul.rs:80-99 -> add_factor_mul_slice_assign (common/slice.rs:61-70): reduce_add(acc, factor * rhs) */
void orc_crt_poly_add_mul_factor_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *acc,
                                        const uint64_t *rhs, const uint64_t *factors) {
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < n; ++j) {
            const uint64_t prod = orc_shoup_mul(factors[2 * i], factors[2 * i + 1], rhs[i * n + j], moduli[i]);
            const uint64_t sum = acc[i * n + j] + prod; /* common/slice.rs:7-12 */
            acc[i * n + j] = sum - moduli[i] < sum ? sum - moduli[i] : sum;
        }
}

static void rotate_right_words(uint64_t *poly, size_t n, size_t r, uint64_t *tmp) {
    /* slice::rotate_right(r): element i moves to (i + r) mod n */
    for (size_t i = 0; i < n; ++i) tmp[(i + r) % n] = poly[i];
    memcpy(poly, tmp, n * sizeof(uint64_t));
}

/* crt/mul.rs:102-127 (= CrtGlwe::mul_monic_monomial_assign, glwe/crt.rs:76-113, per CRT polynomial):
 * r < n: rotate_right(r), negate poly[0..r];  n <= r < 2n: rotate_right(r - n), negate poly[r - n..]. */
int orc_crt_poly_mul_monomial_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *data, size_t r) {
    if (r >= 2 * n) return 1;
    uint64_t *tmp = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!tmp) return 1;
    for (size_t i = 0; i < L; ++i) {
        uint64_t *poly = data + i * n;
        if (r < n) {
            rotate_right_words(poly, n, r, tmp);
            for (size_t j = 0; j < r; ++j) poly[j] = orc_reduce_neg(moduli[i], poly[j]);
        } else {
            const size_t rr = r - n;
            rotate_right_words(poly, n, rr, tmp);
            for (size_t j = rr; j < n; ++j) poly[j] = orc_reduce_neg(moduli[i], poly[j]);
        }
    }
    free(tmp);
    return 0;
}

/* dcrt/inv.rs:55-68 -> BarrettModulus::reduce_inv_slice_to (primus_modulus/src/barrett/slice.rs:535-557):
 * Montgomery batch inversion over each limb polynomial, `out` as the prefix-product buffer.
 * Returns 1 where the reference panics (total product not invertible). */
int orc_dcrt_poly_inv_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, uint64_t *out) {
    for (size_t i = 0; i < L; ++i) {
        orc_barrett m;
        if (orc_barrett_new(moduli[i], &m)) return 1;
        const uint64_t *in = a + i * n;
        uint64_t *o = out + i * n;
        if (n == 0) continue;
        uint64_t total = 1;
        for (size_t j = 0; j < n; ++j) {
            o[j] = total;
            total = orc_barrett_mul(&m, total, in[j]);
        }
        uint64_t suffix = orc_inv_mod(total, moduli[i]);
        if (suffix == 0) return 1;
        for (size_t j = n; j-- > 0;) {
            o[j] = orc_barrett_mul(&m, o[j], suffix);
            suffix = orc_barrett_mul(&m, suffix, in[j]);
        }
    }
    return 0;
}

/* primus_poly/src/poly/mul.rs:107-134 (output zeroed first: the reference accumulates into a
 * caller-zeroed buffer) */
void orc_naive_negacyclic_mul(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n) {
    orc_barrett m;
    if (orc_barrett_new(q, &m)) return;
    memset(c, 0, n * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j <= i; ++j) c[i] = orc_barrett_mul_add(&m, a[j], b[i - j], c[i]);
    for (size_t i = n; i < 2 * n - 1; ++i) {
        size_t k = i - n;
        for (size_t j = i - n + 1; j < n; ++j) c[k] = orc_reduce_sub(q, c[k], orc_barrett_mul(&m, a[j], b[i - j]));
    }
}

/* ========================================================================== */
/* RNSBase<u64, BarrettModulus> — primus_rns/src/base.rs                        */
/* ========================================================================== */

struct orc_rns {
    size_t count, value_len;
    uint64_t *moduli;
    uint64_t *product;   /* value_len */
    uint64_t *punctured; /* count * value_len */
    shoup_t *inv_punct;  /* (Q/q_i)^-1 mod q_i */
};

static uint64_t gcd64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }

/* big_integer.rs: mul_value_assign */
static uint64_t big_mul_value_assign(uint64_t *x, size_t len, uint64_t v) {
    uint64_t carry = 0;
    for (size_t i = 0; i < len; ++i) {
        u128 p = (u128)x[i] * v + carry;
        x[i] = (uint64_t)p; carry = (uint64_t)(p >> 64);
    }
    return carry;
}

/* big_integer.rs:282-298 mul_value_add_to: acc += self * value, returns carry */
static uint64_t big_mul_value_add_to(const uint64_t *self, size_t len, uint64_t v, uint64_t *acc) {
    if (v == 0) return 0;
    uint64_t carry = 0;
    for (size_t i = 0; i < len; ++i) {
        u128 p = (u128)self[i] * v + acc[i] + carry; /* carrying_mul_add */
        acc[i] = (uint64_t)p; carry = (uint64_t)(p >> 64);
    }
    return carry;
}

static int big_cmp(const uint64_t *a, const uint64_t *b, size_t len) { /* big_integer.rs:342-356 */
    for (size_t i = len; i-- > 0;) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
    return 0;
}

static int big_sub_assign(uint64_t *a, const uint64_t *b, size_t len) {
    uint64_t borrow = 0;
    for (size_t i = 0; i < len; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return (int)borrow;
}

static int big_add_assign(uint64_t *a, const uint64_t *b, size_t len) {
    uint64_t carry = 0;
    for (size_t i = 0; i < len; ++i) {
        u128 s = (u128)a[i] + b[i] + carry;
        a[i] = (uint64_t)s; carry = (uint64_t)(s >> 64);
    }
    return (int)carry;
}

static uint64_t big_mod_u64(const uint64_t *a, size_t len, uint64_t q) {
    u128 r = 0;
    for (size_t i = len; i-- > 0;) r = ((r << 64) | a[i]) % q;
    return (uint64_t)r;
}

/* returns carry-out (value shifted out of the top limb must be zero for the callers) */
static uint64_t big_shl_assign(uint64_t *a, size_t len, uint32_t bits) {
    uint64_t out = 0;
    while (bits >= 64) {
        out |= a[len - 1];
        for (size_t i = len - 1; i > 0; --i) a[i] = a[i - 1];
        a[0] = 0; bits -= 64;
    }
    if (bits) {
        out |= a[len - 1] >> (64 - bits);
        for (size_t i = len - 1; i > 0; --i) a[i] = (a[i] << bits) | (a[i - 1] >> (64 - bits));
        a[0] <<= bits;
    }
    return out;
}

/* base.rs:79-117 */
int orc_rns_new(const uint64_t *moduli, size_t count, orc_rns **out) {
    if (count == 0) return ORC_ERR_EMPTY_BASE; /* :47-49 */
    for (size_t i = 0; i < count; ++i)
        for (size_t j = i + 1; j < count; ++j)
            if (gcd64(moduli[i], moduli[j]) != 1) return ORC_ERR_COPRIME; /* :83-89 */
    orc_rns *b = (orc_rns *)calloc(1, sizeof(*b));
    b->count = count;
    b->moduli = (uint64_t *)malloc(count * sizeof(uint64_t));
    memcpy(b->moduli, moduli, count * sizeof(uint64_t));
    /* multiply_many_values (big_integer.rs:675-686): length grows only when a carry appears */
    uint64_t *prod = (uint64_t *)calloc(count, sizeof(uint64_t));
    size_t len = 1; prod[0] = moduli[0];
    for (size_t i = 1; i < count; ++i) {
        uint64_t c = big_mul_value_assign(prod, len, moduli[i]);
        if (c) prod[len++] = c;
    }
    b->value_len = len; b->product = prod;
    /* multiply_many_values_except_to (:713-731), zero padded to value_len */
    b->punctured = (uint64_t *)calloc(count * len, sizeof(uint64_t));
    for (size_t i = 0; i < count; ++i) {
        uint64_t *p = b->punctured + i * len;
        p[0] = 1; size_t l = 1;
        for (size_t j = 0; j < count; ++j) {
            if (j == i) continue;
            uint64_t c = big_mul_value_assign(p, l, moduli[j]);
            if (c) p[l++] = c;
        }
    }
    /* :98-109 */
    b->inv_punct = (shoup_t *)malloc(count * sizeof(shoup_t));
    for (size_t i = 0; i < count; ++i) {
        uint64_t r = big_mod_u64(b->punctured + i * len, len, moduli[i]);
        b->inv_punct[i] = shoup_new(orc_inv_mod(r, moduli[i]), moduli[i]);
    }
    *out = b;
    return ORC_OK;
}

void orc_rns_free(orc_rns *b) {
    if (!b) return;
    free(b->moduli); free(b->product); free(b->punctured); free(b->inv_punct); free(b);
}
size_t orc_rns_moduli_count(const orc_rns *b) { return b->count; }
size_t orc_rns_value_len(const orc_rns *b) { return b->value_len; }
const uint64_t *orc_rns_moduli_product(const orc_rns *b) { return b->product; }
const uint64_t *orc_rns_punctured_product(const orc_rns *b) { return b->punctured; }

/* base.rs:609-633 */
void orc_rns_compose_to(const orc_rns *b, const uint64_t *residues, uint64_t *value) {
    const size_t len = b->value_len;
    memset(value, 0, len * sizeof(uint64_t));
    for (size_t i = 0; i < b->count; ++i) {
        uint64_t product = shoup_full(b->inv_punct[i], residues[i], b->moduli[i]);
        uint64_t carry = big_mul_value_add_to(b->punctured + i * len, len, product, value);
        if (carry != 0 || big_cmp(value, b->product, len) >= 0) (void)big_sub_assign(value, b->product, len);
    }
}

/* base.rs:648-675: gather residue i of coefficient c from multi_residues[i*count + c] */
void orc_rns_compose_multiple_values_to(const orc_rns *b, const uint64_t *multi_residues,
                                        uint64_t *big_uint_values, size_t value_count) {
    uint64_t scratch[64];
    for (size_t c = 0; c < value_count; ++c) {
        for (size_t i = 0; i < b->count; ++i) scratch[i] = multi_residues[i * value_count + c];
        orc_rns_compose_to(b, scratch, big_uint_values + c * b->value_len);
    }
}

/* base.rs:235-246 decompose_to: value mod q_i */
void orc_rns_decompose_to(const orc_rns *b, const uint64_t *value, uint64_t *residues) {
    for (size_t i = 0; i < b->count; ++i) residues[i] = big_mod_u64(value, b->value_len, b->moduli[i]);
}

/* base.rs:457-481 */
void orc_rns_decompose_big_uint_values_to(const orc_rns *b, const uint64_t *big_uint_values,
                                          uint64_t *multi_residues, size_t value_count) {
    for (size_t i = 0; i < b->count; ++i)
        for (size_t c = 0; c < value_count; ++c)
            multi_residues[i * value_count + c] = big_mod_u64(big_uint_values + c * b->value_len, b->value_len, b->moduli[i]);
}

/* ========================================================================== */
/* BaseConverter — primus_rns/src/converter.rs                                  */
/* ========================================================================== */

struct orc_conv {
    const orc_rns *in, *out;   /* borrowed: the caller keeps both bases alive */
    orc_barrett *out_mod;      /* BarrettModulus of every output modulus */
    uint64_t *matrix;          /* out.count rows x in.count columns: (Q/q_i) mod p_j  (converter.rs:54-62) */
    uint64_t q_mod_p0;         /* Q mod p_0 (exact_convert_array, converter.rs:345) */
};

/* converter.rs:43-69 */
int orc_conv_new(const orc_rns *in, const orc_rns *out, orc_conv **res) {
    orc_conv *c = (orc_conv *)calloc(1, sizeof(*c));
    c->in = in; c->out = out;
    c->matrix = (uint64_t *)malloc(in->count * out->count * sizeof(uint64_t));
    c->out_mod = (orc_barrett *)malloc(out->count * sizeof(orc_barrett));
    for (size_t j = 0; j < out->count; ++j) {
        orc_barrett_new(out->moduli[j], &c->out_mod[j]);
        for (size_t i = 0; i < in->count; ++i)
            c->matrix[j * in->count + i] = big_mod_u64(in->punctured + i * in->value_len, in->value_len, out->moduli[j]);
    }
    c->q_mod_p0 = big_mod_u64(in->product, in->value_len, out->moduli[0]);
    *res = c;
    return ORC_OK;
}

void orc_conv_free(orc_conv *c) { if (c) { free(c->matrix); free(c->out_mod); free(c); } }
const uint64_t *orc_conv_matrix(const orc_conv *c) { return c->matrix; }

/* compact/slice.rs:369-405 (reduce_dot_product): 128-bit accumulation in chunks of
 * DOT_PRODUCT_INNER_CHUNK = 16 (compact/mod.rs:14), BarrettModulus::reduce of each chunk
 * (barrett/mod.rs:99-139), chunks folded with reduce_add. */
static uint64_t dot_product_mod(const orc_barrett *m, const uint64_t *a, const uint64_t *b, size_t len) {
    const size_t K = 16;
    uint64_t inter = 0;
    size_t full = len / K;
    for (size_t ch = 0; ch < full; ++ch) {
        u128 c = 0;
        for (size_t t = 0; t < K; ++t) c += (u128)a[ch * K + t] * b[ch * K + t];
        inter = orc_reduce_add(m->value, inter, orc_barrett_reduce_wide(m, (uint64_t)c, (uint64_t)(c >> 64)));
    }
    u128 c = 0;
    for (size_t t = full * K; t < len; ++t) c += (u128)a[t] * b[t];
    return orc_reduce_add(m->value, orc_barrett_reduce_wide(m, (uint64_t)c, (uint64_t)(c >> 64)), inter);
}

/* converter.rs:111-141 */
void orc_conv_fast_convert(const orc_conv *c, const uint64_t *residues_in, uint64_t *residues_out, uint64_t *scratch) {
    const orc_rns *in = c->in;
    for (size_t i = 0; i < in->count; ++i) scratch[i] = shoup_full(in->inv_punct[i], residues_in[i], in->moduli[i]);
    for (size_t j = 0; j < c->out->count; ++j)
        residues_out[j] = dot_product_mod(&c->out_mod[j], scratch, c->matrix + j * in->count, in->count);
}

/* converter.rs:144-178: coefficient-major scratch; the `inv == 1` arm (x mod q_i) equals the
 * Shoup product by 1 */
static void fill_scratch(const orc_conv *c, const uint64_t *crt_poly_in, size_t poly_length, uint64_t *scratch) {
    const orc_rns *in = c->in;
    for (size_t i = 0; i < in->count; ++i)
        for (size_t t = 0; t < poly_length; ++t) {
            const uint64_t x = crt_poly_in[i * poly_length + t];
            scratch[t * in->count + i] = in->inv_punct[i].value == 1 ? x % in->moduli[i]
                                                                     : shoup_full(in->inv_punct[i], x, in->moduli[i]);
        }
}

/* converter.rs:192-218 */
void orc_conv_fast_convert_array(const orc_conv *c, const uint64_t *crt_poly_in, uint64_t *crt_poly_out,
                                 size_t poly_length, uint64_t *scratch) {
    const size_t lin = c->in->count;
    fill_scratch(c, crt_poly_in, poly_length, scratch);
    for (size_t j = 0; j < c->out->count; ++j)
        for (size_t t = 0; t < poly_length; ++t)
            crt_poly_out[j * poly_length + t] = dot_product_mod(&c->out_mod[j], scratch + t * lin, c->matrix + j * lin, lin);
}

/* converter.rs:274-364: the only floating-point code near the path.  v_i = f64(temp_i) / f64(q_i)
 * (IEEE division), summed left to right from 0.0, rounded by (sum + 0.5) as u64 (truncation,
 * saturating).  Compile without -ffast-math / FMA contraction (there is no a*b+c here). */
int orc_conv_exact_convert_array(const orc_conv *c, const uint64_t *crt_poly_in, uint64_t *crt_poly_out,
                                 size_t poly_length) {
    if (c->out->count != 1) return ORC_ERR_BAD_ARG; /* assert_eq!(output_moduli_count(), 1) */
    const orc_rns *in = c->in;
    const size_t lin = in->count;
    uint64_t *temp = (uint64_t *)malloc(lin * poly_length * sizeof(uint64_t));
    fill_scratch(c, crt_poly_in, poly_length, temp);
    const orc_barrett *p = &c->out_mod[0];
    for (size_t t = 0; t < poly_length; ++t) {
        volatile double sum = 0.0; /* Iterator::sum::<f64>() starts at 0.0 and adds in order */
        for (size_t i = 0; i < lin; ++i) {
            const double dividend = (double)temp[t * lin + i];
            const double divisor = (double)in->moduli[i];
            sum = sum + dividend / divisor;
        }
        const double r = sum + 0.5;
        uint64_t v;
        if (!(r > 0.0)) v = 0; else if (r >= 18446744073709551616.0) v = UINT64_MAX; else v = (uint64_t)r;
        const uint64_t dot = dot_product_mod(p, temp + t * lin, c->matrix, lin);
        const uint64_t vq = orc_barrett_mul(p, v, c->q_mod_p0);
        crt_poly_out[t] = orc_reduce_sub(p->value, dot, vq);
    }
    free(temp);
    return ORC_OK;
}

/* base.rs:279-312 + slice::wrapping_decompose_chunk_to :721-730 */
void orc_rns_wrapping_decompose_small_values_to(const orc_rns *b, const uint64_t *small_values,
                                                uint64_t *multi_residues, size_t value_count,
                                                uint64_t small_value_modulus) {
    if (small_value_modulus != 2) {
        uint64_t half = (small_value_modulus + 1) / 2;
        for (size_t i = 0; i < b->count; ++i) {
            uint64_t temp = b->moduli[i] - small_value_modulus;
            uint64_t *res = multi_residues + i * value_count;
            for (size_t c = 0; c < value_count; ++c) {
                uint64_t v = small_values[c];
                res[c] = v < half ? v : temp + v;
            }
        }
    } else {
        for (size_t i = 0; i < b->count; ++i) memcpy(multi_residues + i * value_count, small_values, value_count * sizeof(uint64_t));
    }
}

/* base.rs:326-384 + slice::wrapping_decompose_chunk_scaled_to :739-757: centred lift, Shoup product by the
 * modulus' factor, compact reduce_add into acc; small_value_modulus == 2 takes add_factor_mul_slice_assign on the
 * raw values (:371-378).  `factors` = count (value, quotient) pairs. */
void orc_rns_add_wrapping_decompose_small_values_scaled(const orc_rns *b, const uint64_t *small_values, uint64_t *acc,
                                                        size_t value_count, uint64_t small_value_modulus,
                                                        const uint64_t *factors) {
    const uint64_t half = (small_value_modulus + 1) / 2;
    for (size_t i = 0; i < b->count; ++i) {
        const uint64_t q = b->moduli[i], temp = q - small_value_modulus;
        uint64_t *a = acc + i * value_count;
        for (size_t c = 0; c < value_count; ++c) {
            const uint64_t v = small_values[c];
            const uint64_t centred = (small_value_modulus != 2 && v >= half) ? temp + v : v;
            a[c] = orc_reduce_add(q, a[c], orc_shoup_mul(factors[2 * i], factors[2 * i + 1], centred, q));
        }
    }
}

/* base.rs:398-416 (and add_decompose_small_polynomial_scaled :429-443): the same without the lift */
void orc_rns_add_decompose_small_values_scaled(const orc_rns *b, const uint64_t *small_values, uint64_t *acc,
                                               size_t value_count, const uint64_t *factors) {
    for (size_t i = 0; i < b->count; ++i) {
        const uint64_t q = b->moduli[i];
        uint64_t *a = acc + i * value_count;
        for (size_t c = 0; c < value_count; ++c)
            a[c] = orc_reduce_add(q, a[c], orc_shoup_mul(factors[2 * i], factors[2 * i + 1], small_values[c], q));
    }
}

/* ========================================================================== */
/* BigUintApproxSignedBasis<u64> — primus_decompose/src/big_integer/{basis,common}.rs */
/* ========================================================================== */

typedef struct { uint64_t mask; size_t index; uint32_t shr_bits; uint32_t shl_bits; /* 0 = None */ } value_mask_t;

struct orc_basis {
    size_t value_len, moduli_count, decompose_length;
    uint32_t log_basis, drop_bits;
    uint64_t basis, basis_minus_one, carry_mask;
    int mode; /* 0 Plain, 1 CarryOnly, 2 AdjustOnly, 3 AdjustAndCarry */
    uint64_t *threshold, *add; /* value_len each (zero when absent) */
    size_t carry_index; uint64_t carry_bit_mask;
    uint64_t *scalars, *scalars_residue;
    value_mask_t *masks;
    uint64_t *modulus_sub_basis; /* Q - B, value_len limbs (basis.rs:133-134) */
};

/* common.rs:83-103 */
static value_mask_t value_mask_new(uint64_t mask, uint32_t drop_bits) {
    value_mask_t v; v.mask = mask; v.index = drop_bits / 64; v.shr_bits = drop_bits % 64;
    uint32_t lz = mask ? (uint32_t)__builtin_clzll(mask) : 64;
    v.shl_bits = lz < v.shr_bits ? 64 - v.shr_bits : 0;
    return v;
}
/* common.rs:107-124 */
static value_mask_t value_mask_next(value_mask_t v, uint32_t advance) {
    uint32_t shr = advance + v.shr_bits;
    if (shr >= 64) { v.index += 1; shr -= 64; }
    v.shr_bits = shr;
    uint32_t lz = v.mask ? (uint32_t)__builtin_clzll(v.mask) : 64;
    v.shl_bits = lz < shr ? 64 - shr : 0;
    return v;
}
/* common.rs:132-140 */
static uint64_t value_mask_get(const value_mask_t *v, const uint64_t *value) {
    uint64_t lower = value[v->index] >> v->shr_bits;
    if (v->shl_bits) return (lower | (value[v->index + 1] << v->shl_bits)) & v->mask;
    return lower & v->mask;
}

/* basis.rs:40-211 */
int orc_basis_new(const orc_rns *rns, uint32_t log_basis, size_t reverse_length, orc_basis **out) {
    const size_t len = rns->value_len;
    const uint64_t *modulus = rns->product;
    if (modulus[len - 1] == 0 || log_basis == 0 || log_basis >= 64) return ORC_ERR_BAD_ARG; /* :50-51 */
    uint32_t unused_bits = (uint32_t)__builtin_clzll(modulus[len - 1]);
    uint64_t basis = 1ull << log_basis, bm1 = basis - 1;
    uint32_t bits = 64 * (uint32_t)len - unused_bits;
    size_t dlen = bits / log_basis;
    uint32_t drop = bits - (uint32_t)dlen * log_basis;
    if (reverse_length) { /* :63-68 */
        if (dlen < reverse_length) return ORC_ERR_BAD_ARG;
        dlen = reverse_length;
        drop = bits - (uint32_t)reverse_length * log_basis;
    }
    if (dlen == 0) return ORC_ERR_BAD_ARG;

    orc_basis *b = (orc_basis *)calloc(1, sizeof(*b));
    b->value_len = len; b->moduli_count = rns->count; b->decompose_length = dlen;
    b->log_basis = log_basis; b->drop_bits = drop; b->basis = basis; b->basis_minus_one = bm1;
    int has_carry = drop > 0;
    if (has_carry) { uint32_t cb = drop - 1; b->carry_index = cb / 64; b->carry_bit_mask = 1ull << (cb % 64); } /* :72-79 */
    b->carry_mask = log_basis == 1 ? (1ull << 1) : ((1ull << log_basis) | (1ull << (log_basis - 1))); /* :81-85 */

    /* split value :87-131 */
    uint64_t *split = (uint64_t *)calloc(len, sizeof(uint64_t));
    int has_split = 0;
    if (log_basis == 1) {
        if (drop != 0) {
            for (size_t i = 0; i < dlen; ++i) { big_shl_assign(split, len, 1); split[0] |= 1; }
            big_shl_assign(split, len, 1); split[0] |= 1;
            big_shl_assign(split, len, drop - 1);
            has_split = big_cmp(split, modulus, len) < 0;
        }
    } else {
        for (size_t i = 0; i < dlen; ++i) { big_shl_assign(split, len, log_basis); split[0] |= bm1 >> 1; }
        if (drop > 0) {
            big_shl_assign(split, len, 1); split[0] |= 1;
            big_shl_assign(split, len, drop - 1);
        } else {
            uint64_t one[64] = {1};
            big_add_assign(split, one, len);
        }
        has_split = big_cmp(split, modulus, len) < 0;
    }
    b->threshold = split;
    b->add = (uint64_t *)calloc(len, sizeof(uint64_t));
    if (has_split) { /* make_adjust_add :137-147: (2^bits - 1) - (Q - 1) */
        for (size_t i = 0; i < len; ++i) b->add[i] = ~0ull;
        b->add[len - 1] >>= unused_bits;
        uint64_t *qm1 = (uint64_t *)malloc(len * sizeof(uint64_t));
        memcpy(qm1, modulus, len * sizeof(uint64_t));
        uint64_t one[64] = {1};
        big_sub_assign(qm1, one, len);
        big_sub_assign(b->add, qm1, len);
        free(qm1);
    } else {
        memset(split, 0, len * sizeof(uint64_t));
    }
    b->mode = (has_split ? 2 : 0) | (has_carry ? 1 : 0); /* :183-197 */

    /* scalars :149-163: 2^(drop + j*log_basis) */
    b->scalars = (uint64_t *)calloc(dlen * len, sizeof(uint64_t));
    for (size_t j = 0; j < dlen; ++j) {
        uint64_t *s = b->scalars + j * len;
        if (j == 0) { s[0] = 1; big_shl_assign(s, len, drop); }
        else { memcpy(s, s - len, len * sizeof(uint64_t)); big_shl_assign(s, len, log_basis); }
    }
    /* scalars_residue :165-173 */
    b->scalars_residue = (uint64_t *)calloc(dlen * rns->count, sizeof(uint64_t));
    for (size_t j = 0; j < dlen; ++j) orc_rns_decompose_to(rns, b->scalars + j * len, b->scalars_residue + j * rns->count);
    /* value masks :175-181 */
    b->modulus_sub_basis = (uint64_t *)calloc(len, sizeof(uint64_t)); /* basis.rs:133-134 */
    {
        uint64_t bv[64] = {0};
        bv[0] = basis;
        memcpy(b->modulus_sub_basis, modulus, len * sizeof(uint64_t));
        (void)big_sub_assign(b->modulus_sub_basis, bv, len);
    }
    b->masks = (value_mask_t *)malloc(dlen * sizeof(value_mask_t));
    b->masks[0] = value_mask_new(bm1, drop);
    for (size_t j = 1; j < dlen; ++j) b->masks[j] = value_mask_next(b->masks[j - 1], log_basis);
    *out = b;
    return ORC_OK;
}

void orc_basis_free(orc_basis *b) {
    if (!b) return;
    free(b->threshold); free(b->add); free(b->scalars); free(b->scalars_residue); free(b->masks); free(b->modulus_sub_basis); free(b);
}
size_t orc_basis_decompose_length(const orc_basis *b) { return b->decompose_length; }
uint32_t orc_basis_log_basis(const orc_basis *b) { return b->log_basis; }
uint32_t orc_basis_drop_bits(const orc_basis *b) { return b->drop_bits; }
uint64_t orc_basis_basis_value(const orc_basis *b) { return b->basis; }
int orc_basis_init_mode(const orc_basis *b) { return b->mode; }
const uint64_t *orc_basis_threshold(const orc_basis *b) { return b->threshold; }
const uint64_t *orc_basis_adjust_add(const orc_basis *b) { return b->add; }
const uint64_t *orc_basis_scalars(const orc_basis *b) { return b->scalars; }
const uint64_t *orc_basis_scalars_residue(const orc_basis *b) { return b->scalars_residue; }

/* basis.rs:326-367 */
void orc_basis_init_value_carry_slice_inplace(const orc_basis *b, uint64_t *values, uint8_t *carries, size_t count) {
    const size_t len = b->value_len;
    for (size_t c = 0; c < count; ++c) {
        uint64_t *v = values + c * len;
        if (b->mode & 2) {
            if (big_cmp(v, b->threshold, len) >= 0) (void)big_add_assign(v, b->add, len);
        }
        carries[c] = (b->mode & 1) ? (uint8_t)((v[b->carry_index] & b->carry_bit_mask) != 0) : 0;
    }
}

/* common.rs:275-285 (unsigned_decompose_to) over a slice (:309-325) */
void orc_basis_unsigned_decompose_slice_to(const orc_basis *b, size_t level, const uint64_t *values,
                                           uint64_t *digits, uint8_t *carries, size_t count) {
    const value_mask_t *vm = &b->masks[level];
    for (size_t c = 0; c < count; ++c) {
        uint64_t temp = value_mask_get(vm, values + c * b->value_len) + (uint64_t)carries[c];
        carries[c] = (uint8_t)((temp & b->carry_mask) != 0);
        digits[c] = temp & b->basis_minus_one;
    }
}

/* basis.rs:371-420 (init_value_carry_slice_to): the out-of-place form */
void orc_basis_init_value_carry_slice_to(const orc_basis *b, const uint64_t *values, uint64_t *adjusted,
                                         uint8_t *carries, size_t count) {
    memcpy(adjusted, values, count * b->value_len * sizeof(uint64_t));
    orc_basis_init_value_carry_slice_inplace(b, adjusted, carries, count);
}

/* common.rs:255-272 (decompose_to) over a slice (:289-306): the SIGNED digit as a residue modulo Q — a digit
 * temp with the carry set stands for temp - B, stored as (Q - B) + temp; temp == B is the digit 0. */
void orc_basis_decompose_slice_to(const orc_basis *b, size_t level, const uint64_t *values, uint64_t *decomposed,
                                  uint8_t *carries, size_t count) {
    const value_mask_t *vm = &b->masks[level];
    const size_t len = b->value_len;
    for (size_t c = 0; c < count; ++c) {
        uint64_t temp = value_mask_get(vm, values + c * len) + (uint64_t)carries[c];
        uint64_t *d = decomposed + c * len;
        carries[c] = (uint8_t)((temp & b->carry_mask) != 0);
        memset(d, 0, len * sizeof(uint64_t));
        if (carries[c]) {
            if (temp <= b->basis_minus_one) {
                uint64_t tv[64] = {0};
                tv[0] = temp;
                memcpy(d, b->modulus_sub_basis, len * sizeof(uint64_t));
                (void)big_add_assign(d, tv, len);
            }
        } else {
            d[0] = temp;
        }
    }
}

/* ========================================================================== */
/* external product — primus_lattice/src/glwe/{dcrt.rs,crt.rs}                   */
/* ========================================================================== */

/* glwe/dcrt.rs:178-255 */
void orc_add_dcrt_glev_mul_crt_poly_assign(const orc_dcrt *table, const orc_rns *rns,
                                           const orc_basis *basis, size_t k, uint64_t *acc,
                                           const uint64_t *dcrt_glev, const uint64_t *crt_poly) {
    const size_t n = table->n, L = table->count, W = L * n, len = rns->value_len;
    const size_t glwe_len = (k + 1) * W;
    uint64_t *adjust = (uint64_t *)malloc(n * len * sizeof(uint64_t));
    uint64_t *digits = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint8_t *carries = (uint8_t *)malloc(n);
    uint64_t *multi = (uint64_t *)malloc(W * sizeof(uint64_t));

    orc_rns_compose_multiple_values_to(rns, crt_poly, adjust, n);           /* :219-224 */
    orc_basis_init_value_carry_slice_inplace(basis, adjust, carries, n);     /* :226 */
    for (size_t j = 0; j < basis->decompose_length; ++j) {                   /* :228-254 */
        const uint64_t *glwe = dcrt_glev + j * glwe_len;
        orc_basis_unsigned_decompose_slice_to(basis, j, adjust, digits, carries, n);
        orc_rns_wrapping_decompose_small_values_to(rns, digits, multi, n, basis->basis);
        orc_dcrt_transform_slice(table, multi);
        /* add_dcrt_glwe_mul_dcrt_polynomial_assign :108-126 */
        for (size_t c = 0; c <= k; ++c) orc_dcrt_poly_add_mul_assign(table, acc + c * W, glwe + c * W, multi);
    }
    free(adjust); free(digits); free(carries); free(multi);
}

/* glwe/dcrt.rs:258-338 (and glev/dcrt.rs:113-175 when acc starts at zero): as above with the polynomial given as
 * big integers modulo Q — init_value_carry_slice_to replaces compose + init_value_carry_slice_inplace */
void orc_add_dcrt_glev_mul_big_uint_poly_assign(const orc_dcrt *table, const orc_rns *rns, const orc_basis *basis,
                                                size_t k, uint64_t *acc, const uint64_t *dcrt_glev,
                                                const uint64_t *big_uint_poly) {
    const size_t n = table->n, L = table->count, W = L * n, len = rns->value_len;
    const size_t glwe_len = (k + 1) * W;
    uint64_t *adjust = (uint64_t *)malloc(n * len * sizeof(uint64_t));
    uint64_t *digits = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint8_t *carries = (uint8_t *)malloc(n);
    uint64_t *multi = (uint64_t *)malloc(W * sizeof(uint64_t));
    orc_basis_init_value_carry_slice_to(basis, big_uint_poly, adjust, carries, n); /* :301-306 */
    for (size_t j = 0; j < basis->decompose_length; ++j) {                         /* :308-337 */
        const uint64_t *glwe = dcrt_glev + j * glwe_len;
        orc_basis_unsigned_decompose_slice_to(basis, j, adjust, digits, carries, n);
        orc_rns_wrapping_decompose_small_values_to(rns, digits, multi, n, basis->basis);
        orc_dcrt_transform_slice(table, multi);
        for (size_t c = 0; c <= k; ++c) orc_dcrt_poly_add_mul_assign(table, acc + c * W, glwe + c * W, multi);
    }
    free(adjust); free(digits); free(carries); free(multi);
}

/* glwe/crt.rs:200-227 */
void orc_mul_dcrt_ggsw_to(const orc_dcrt *table, const orc_rns *rns, const orc_basis *basis, size_t k,
                          const uint64_t *crt_glwe, const uint64_t *dcrt_ggsw, uint64_t *result) {
    const size_t W = table->count * table->n;
    const size_t glwe_len = (k + 1) * W;
    const size_t glev_len = basis->decompose_length * glwe_len;
    memset(result, 0, glwe_len * sizeof(uint64_t)); /* :217 */
    for (size_t i = 0; i <= k; ++i)
        orc_add_dcrt_glev_mul_crt_poly_assign(table, rns, basis, k, result, dcrt_ggsw + i * glev_len, crt_glwe + i * W);
}
