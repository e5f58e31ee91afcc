/*
 * pfhe_oracle_rns32.c — the <u32> instantiations of the reference's RNS / gadget / external-product generics,
 * restated with T = u32 (limbs, residues and digits are 32-bit words, the widening type is u64):
 *   RNSBase<u32, BarrettModulus<u32>>      primus_rns/src/base.rs:26-117 (generic over T: FheUint)
 *   BaseConverter<u32, BarrettModulus<u32>> primus_rns/src/converter.rs:21-364
 *   BigUintApproxSignedBasis<u32>          primus_decompose/src/big_integer/basis.rs:33 (the type tests/big_uint.rs:13 runs)
 *   CrtGlwe<u32>::mul_dcrt_ggsw_to         primus_lattice/src/glwe/crt.rs:200-227 over U32DcrtTable (dcrt/prime32.rs:11)
 *
 * TEST INFRASTRUCTURE ONLY (see pfhe_oracle.h).  Written against the reference source with 32-bit limb arithmetic
 * throughout — NOT derived from the 64-bit restatement in pfhe_oracle.c — so the GPU's u32 entry points (64-bit
 * arithmetic inside, 32-bit words in memory) are checked against genuinely 32-bit code.
 * PARITY STATUS: unpinned against reference binary output, as the rest of the oracle (no Rust toolchain here).
 */
#include <stdlib.h>
#include <string.h>

#include "pfhe_oracle.h"

typedef uint32_t w32;
typedef uint64_t wide;
#define WBITS 32u
#define MAX_LIMBS 128

/* ---- modular arithmetic on one 32-bit modulus ---- */
typedef struct { w32 value, quotient; } shoup32_t; /* ShoupFactor<u32> (primus_factor/src/shoup_factor/mod.rs:35-41) */

static shoup32_t shoup32_new(w32 v, w32 q) { shoup32_t s = {v, (w32)(((wide)v << 32) / q)}; return s; }
/* lazy_factor_mul_modulo :124-129, factor_mul_modulo :139-143 */
static w32 shoup32_mul(shoup32_t s, w32 b, w32 q) {
    const w32 hw = (w32)(((wide)s.quotient * b) >> 32);
    const w32 t = s.value * b - q * hw;
    const w32 d = t - q;
    return t < d ? t : d;
}
static w32 reduce_add32(w32 q, w32 a, w32 b) { const w32 s = a + b, d = s - q; return s < d ? s : d; } /* compact/primitive.rs:10-22 */
static w32 gcd32(w32 a, w32 b) { while (b) { w32 t = a % b; a = b; b = t; } return a; }
static w32 inv_mod32(w32 a, w32 q) {
    int64_t t = 0, nt = 1, r = q, nr = a % q;
    while (nr) { int64_t k = r / nr, x = t - k * nt; t = nt; nt = x; x = r - k * nr; r = nr; nr = x; }
    if (t < 0) t += q;
    return (w32)t;
}

/* ---- big integers as little-endian u32 limbs (primus_integer/src/big_integer.rs) ---- */
static w32 big32_mul_value_assign(w32 *x, size_t len, w32 v) {
    w32 carry = 0;
    for (size_t i = 0; i < len; ++i) { wide p = (wide)x[i] * v + carry; x[i] = (w32)p; carry = (w32)(p >> 32); }
    return carry;
}
/* :282-298 mul_value_add_to */
static w32 big32_mul_value_add_to(const w32 *self, size_t len, w32 v, w32 *acc) {
    if (v == 0) return 0;
    w32 carry = 0;
    for (size_t i = 0; i < len; ++i) { wide p = (wide)self[i] * v + acc[i] + carry; acc[i] = (w32)p; carry = (w32)(p >> 32); }
    return carry;
}
static int big32_cmp(const w32 *a, const w32 *b, size_t len) { /* :342-356 */
    for (size_t i = len; i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
static int big32_sub_assign(w32 *a, const w32 *b, size_t len) {
    w32 borrow = 0;
    for (size_t i = 0; i < len; ++i) { wide d = (wide)a[i] - b[i] - borrow; a[i] = (w32)d; borrow = (w32)(d >> 32) & 1u; }
    return (int)borrow;
}
static int big32_add_assign(w32 *a, const w32 *b, size_t len) {
    w32 carry = 0;
    for (size_t i = 0; i < len; ++i) { wide s = (wide)a[i] + b[i] + carry; a[i] = (w32)s; carry = (w32)(s >> 32); }
    return (int)carry;
}
static w32 big32_mod(const w32 *a, size_t len, w32 q) {
    wide r = 0;
    for (size_t i = len; i-- > 0;) r = ((r << 32) | a[i]) % q;
    return (w32)r;
}
static w32 big32_shl_assign(w32 *a, size_t len, uint32_t bits) {
    w32 out = 0;
    while (bits >= WBITS) {
        out |= a[len - 1];
        for (size_t i = len - 1; i > 0; --i) a[i] = a[i - 1];
        a[0] = 0; bits -= WBITS;
    }
    if (bits) {
        out |= a[len - 1] >> (WBITS - bits);
        for (size_t i = len - 1; i > 0; --i) a[i] = (a[i] << bits) | (a[i - 1] >> (WBITS - bits));
        a[0] <<= bits;
    }
    return out;
}

/* ========================================================================== */
/* RNSBase<u32, BarrettModulus<u32>> — primus_rns/src/base.rs                   */
/* ========================================================================== */
struct orc_rns32 {
    size_t count, value_len;
    w32 *moduli, *product, *punctured;
    shoup32_t *inv_punct;
};

/* base.rs:79-117 */
int orc_rns32_new(const uint32_t *moduli, size_t count, orc_rns32 **out) {
    if (count == 0) return ORC_ERR_EMPTY_BASE; /* :47-49 */
    for (size_t i = 0; i < count; ++i) /* BarrettModulus::<u32>::new, barrett/mod.rs:39-44: 1 < q, leading_zeros > 1 */
        if (moduli[i] <= 1 || moduli[i] >= (1u << 30)) return ORC_ERR_MODULUS_TOO_LARGE;
    for (size_t i = 0; i < count; ++i)
        for (size_t j = i + 1; j < count; ++j)
            if (gcd32(moduli[i], moduli[j]) != 1) return ORC_ERR_COPRIME; /* :83-89 */
    if (count > MAX_LIMBS) return ORC_ERR_BAD_ARG;
    orc_rns32 *b = (orc_rns32 *)calloc(1, sizeof(*b));
    b->count = count;
    b->moduli = (w32 *)malloc(count * sizeof(w32));
    memcpy(b->moduli, moduli, count * sizeof(w32));
    /* multiply_many_values (big_integer.rs:675-686): a limb is appended only when a carry appears */
    w32 *prod = (w32 *)calloc(count, sizeof(w32));
    size_t len = 1; prod[0] = moduli[0];
    for (size_t i = 1; i < count; ++i) { w32 c = big32_mul_value_assign(prod, len, moduli[i]); if (c) prod[len++] = c; }
    b->value_len = len; b->product = prod;
    /* multiply_many_values_except_to (:713-731), zero padded to value_len */
    b->punctured = (w32 *)calloc(count * len, sizeof(w32));
    for (size_t i = 0; i < count; ++i) {
        w32 *p = b->punctured + i * len;
        p[0] = 1; size_t l = 1;
        for (size_t j = 0; j < count; ++j) {
            if (j == i) continue;
            w32 c = big32_mul_value_assign(p, l, moduli[j]);
            if (c) p[l++] = c;
        }
    }
    b->inv_punct = (shoup32_t *)malloc(count * sizeof(shoup32_t)); /* :98-109 */
    for (size_t i = 0; i < count; ++i)
        b->inv_punct[i] = shoup32_new(inv_mod32(big32_mod(b->punctured + i * len, len, moduli[i]), moduli[i]), moduli[i]);
    *out = b;
    return ORC_OK;
}
void orc_rns32_free(orc_rns32 *b) {
    if (!b) return;
    free(b->moduli); free(b->product); free(b->punctured); free(b->inv_punct); free(b);
}
size_t orc_rns32_moduli_count(const orc_rns32 *b) { return b->count; }
size_t orc_rns32_value_len(const orc_rns32 *b) { return b->value_len; }
const uint32_t *orc_rns32_moduli_product(const orc_rns32 *b) { return b->product; }
const uint32_t *orc_rns32_punctured_product(const orc_rns32 *b) { return b->punctured; }

/* base.rs:609-633 */
void orc_rns32_compose_to(const orc_rns32 *b, const uint32_t *residues, uint32_t *value) {
    const size_t len = b->value_len;
    memset(value, 0, len * sizeof(w32));
    for (size_t i = 0; i < b->count; ++i) {
        const w32 product = shoup32_mul(b->inv_punct[i], residues[i], b->moduli[i]);
        const w32 carry = big32_mul_value_add_to(b->punctured + i * len, len, product, value);
        if (carry != 0 || big32_cmp(value, b->product, len) >= 0) (void)big32_sub_assign(value, b->product, len);
    }
}
/* base.rs:648-675 */
void orc_rns32_compose_multiple_values_to(const orc_rns32 *b, const uint32_t *multi_residues, uint32_t *big_uint_values,
                                          size_t value_count) {
    w32 scratch[MAX_LIMBS];
    for (size_t c = 0; c < value_count; ++c) {
        for (size_t i = 0; i < b->count; ++i) scratch[i] = multi_residues[i * value_count + c];
        orc_rns32_compose_to(b, scratch, big_uint_values + c * b->value_len);
    }
}
/* base.rs:235-246 */
void orc_rns32_decompose_to(const orc_rns32 *b, const uint32_t *value, uint32_t *residues) {
    for (size_t i = 0; i < b->count; ++i) residues[i] = big32_mod(value, b->value_len, b->moduli[i]);
}
/* base.rs:457-481 */
void orc_rns32_decompose_big_uint_values_to(const orc_rns32 *b, const uint32_t *big_uint_values, uint32_t *multi_residues,
                                            size_t value_count) {
    for (size_t i = 0; i < b->count; ++i)
        for (size_t c = 0; c < value_count; ++c)
            multi_residues[i * value_count + c] = big32_mod(big_uint_values + c * b->value_len, b->value_len, b->moduli[i]);
}
/* base.rs:279-312 + slice::wrapping_decompose_chunk_to :721-730 */
void orc_rns32_wrapping_decompose_small_values_to(const orc_rns32 *b, const uint32_t *small_values, uint32_t *multi_residues,
                                                  size_t value_count, uint32_t small_value_modulus) {
    if (small_value_modulus != 2) {
        const w32 half = (small_value_modulus + 1) / 2;
        for (size_t i = 0; i < b->count; ++i) {
            const w32 temp = b->moduli[i] - small_value_modulus;
            w32 *res = multi_residues + i * value_count;
            for (size_t c = 0; c < value_count; ++c) { const w32 v = small_values[c]; res[c] = v < half ? v : temp + v; }
        }
    } else {
        for (size_t i = 0; i < b->count; ++i) memcpy(multi_residues + i * value_count, small_values, value_count * sizeof(w32));
    }
}
/* base.rs:326-384 (+ :739-757); `factors` = count (value, quotient) pairs of ShoupFactor<u32> */
void orc_rns32_add_wrapping_decompose_small_values_scaled(const orc_rns32 *b, const uint32_t *small_values, uint32_t *acc,
                                                          size_t value_count, uint32_t small_value_modulus,
                                                          const uint32_t *factors) {
    const w32 half = (small_value_modulus + 1) / 2;
    for (size_t i = 0; i < b->count; ++i) {
        const w32 q = b->moduli[i], temp = q - small_value_modulus;
        const shoup32_t f = {factors[2 * i], factors[2 * i + 1]};
        w32 *a = acc + i * value_count;
        for (size_t c = 0; c < value_count; ++c) {
            const w32 v = small_values[c];
            const w32 centred = (small_value_modulus != 2 && v >= half) ? temp + v : v;
            a[c] = reduce_add32(q, a[c], shoup32_mul(f, centred, q));
        }
    }
}
/* base.rs:398-416 */
void orc_rns32_add_decompose_small_values_scaled(const orc_rns32 *b, const uint32_t *small_values, uint32_t *acc,
                                                 size_t value_count, const uint32_t *factors) {
    for (size_t i = 0; i < b->count; ++i) {
        const w32 q = b->moduli[i];
        const shoup32_t f = {factors[2 * i], factors[2 * i + 1]};
        w32 *a = acc + i * value_count;
        for (size_t c = 0; c < value_count; ++c) a[c] = reduce_add32(q, a[c], shoup32_mul(f, small_values[c], q));
    }
}

/* ========================================================================== */
/* BaseConverter<u32, BarrettModulus<u32>> — primus_rns/src/converter.rs          */
/* ========================================================================== */
/* BarrettModulus<u32> (primus_modulus/src/barrett/mod.rs:25-31): ratio = floor(2^64 / value) as two 32-bit words
 * (new_unchecked :57-64) */
typedef struct { w32 value, ratio[2]; } barrett32_t;
static barrett32_t barrett32_new(w32 value) {
    /* floor(2^64 / value): 2^64 = value * floor((2^64 - 1) / value) + r, r + 1 <= value; equal only when value | 2^64 */
    wide q = ~(wide)0 / value;
    if ((~(wide)0 % value) + 1 == value) ++q;
    barrett32_t m = {value, {(w32)q, (w32)(q >> 32)}};
    return m;
}
/* lazy_reduce_wide :99-133 + reduce_once (reduce_wide :137-139), every step in 32-bit words */
static w32 barrett32_reduce_wide(const barrett32_t *m, w32 lo, w32 hi) {
    const w32 ah = (w32)(((wide)lo * m->ratio[0]) >> 32);     /* widening_mul_hw */
    const wide b = (wide)lo * m->ratio[1] + ah;               /* carrying_mul(ratio[1], ah) */
    const wide c = (wide)hi * m->ratio[0];                    /* widening_mul */
    const w32 d = hi * m->ratio[1];                           /* wrapping_mul */
    const w32 b0 = (w32)b, b1 = (w32)(b >> 32), c0 = (w32)c, c1 = (w32)(c >> 32);
    const w32 carry = (w32)(b0 + c0) < b0;                    /* overflowing_add(..).1 */
    const w32 bch = b1 + c1 + carry;                          /* carrying_add(..).0 */
    const w32 q = d + bch;
    const w32 r = lo - q * m->value;
    return r >= m->value ? r - m->value : r;
}
/* reduce_dot_product (primus_modulus/src/common/compact/slice.rs:380-405): a [u32; 2] accumulator per chunk of
 * DOT_PRODUCT_INNER_CHUNK = 16 terms (multiply_add: widening product added with carry, overflow of the upper word
 * discarded), reduce, fold with reduce_add; then the remainder */
static w32 dot_product_mod32(const barrett32_t *m, const w32 *a, const w32 *b, size_t len) {
    const size_t K = 16;
    w32 inter = 0;
    const size_t full = len / K;
    for (size_t ch = 0; ch < full; ++ch) {
        wide c = 0;
        for (size_t t = 0; t < K; ++t) c += (wide)a[ch * K + t] * b[ch * K + t];
        inter = reduce_add32(m->value, inter, barrett32_reduce_wide(m, (w32)c, (w32)(c >> 32)));
    }
    wide c = 0;
    for (size_t t = full * K; t < len; ++t) c += (wide)a[t] * b[t];
    return reduce_add32(m->value, barrett32_reduce_wide(m, (w32)c, (w32)(c >> 32)), inter);
}

struct orc_conv32 {
    const orc_rns32 *in, *out; /* borrowed */
    barrett32_t *out_mod;
    w32 *matrix;               /* out.count rows x in.count columns: (Q/q_i) mod p_j (converter.rs:54-62) */
    w32 q_mod_p0;              /* Q mod p_0 (converter.rs:345) */
};

/* converter.rs:43-69 */
int orc_conv32_new(const orc_rns32 *in, const orc_rns32 *out, orc_conv32 **res) {
    orc_conv32 *c = (orc_conv32 *)calloc(1, sizeof(*c));
    c->in = in; c->out = out;
    c->matrix = (w32 *)malloc(in->count * out->count * sizeof(w32));
    c->out_mod = (barrett32_t *)malloc(out->count * sizeof(barrett32_t));
    for (size_t j = 0; j < out->count; ++j) {
        c->out_mod[j] = barrett32_new(out->moduli[j]);
        for (size_t i = 0; i < in->count; ++i)
            c->matrix[j * in->count + i] = big32_mod(in->punctured + i * in->value_len, in->value_len, out->moduli[j]);
    }
    c->q_mod_p0 = big32_mod(in->product, in->value_len, out->moduli[0]);
    *res = c;
    return ORC_OK;
}
void orc_conv32_free(orc_conv32 *c) { if (c) { free(c->matrix); free(c->out_mod); free(c); } }
const uint32_t *orc_conv32_matrix(const orc_conv32 *c) { return c->matrix; }

/* converter.rs:144-178: coefficient-major scratch; `inv == 1` takes x mod q_i */
static void fill_scratch32(const orc_conv32 *c, const w32 *crt_poly_in, size_t poly_length, w32 *scratch) {
    const orc_rns32 *in = c->in;
    for (size_t i = 0; i < in->count; ++i)
        for (size_t t = 0; t < poly_length; ++t) {
            const w32 x = crt_poly_in[i * poly_length + t];
            scratch[t * in->count + i] = in->inv_punct[i].value == 1 ? x % in->moduli[i]
                                                                     : shoup32_mul(in->inv_punct[i], x, in->moduli[i]);
        }
}

/* converter.rs:192-218 */
void orc_conv32_fast_convert_array(const orc_conv32 *c, const uint32_t *crt_poly_in, uint32_t *crt_poly_out,
                                   size_t poly_length, uint32_t *scratch) {
    const size_t lin = c->in->count;
    fill_scratch32(c, crt_poly_in, poly_length, scratch);
    for (size_t j = 0; j < c->out->count; ++j)
        for (size_t t = 0; t < poly_length; ++t)
            crt_poly_out[j * poly_length + t] = dot_product_mod32(&c->out_mod[j], scratch + t * lin, c->matrix + j * lin, lin);
}

/* converter.rs:274-364 with T = u32: v_i = f64(temp_i) / f64(q_i), summed left to right from 0.0, (sum + 0.5) as u32
 * (truncation, saturating); reduce_mul(v, Q mod p) = reduce(widening_mul) (barrett/ops.rs:280-282) */
int orc_conv32_exact_convert_array(const orc_conv32 *c, const uint32_t *crt_poly_in, uint32_t *crt_poly_out,
                                   size_t poly_length) {
    if (c->out->count != 1) return ORC_ERR_BAD_ARG;
    const orc_rns32 *in = c->in;
    const size_t lin = in->count;
    w32 *temp = (w32 *)malloc(lin * poly_length * sizeof(w32));
    fill_scratch32(c, crt_poly_in, poly_length, temp);
    const barrett32_t *p = &c->out_mod[0];
    for (size_t t = 0; t < poly_length; ++t) {
        volatile double sum = 0.0;
        for (size_t i = 0; i < lin; ++i) {
            const double dividend = (double)temp[t * lin + i];
            const double divisor = (double)in->moduli[i];
            sum = sum + dividend / divisor;
        }
        const double r = sum + 0.5;
        w32 v;
        if (!(r > 0.0)) v = 0; else if (r >= 4294967296.0) v = UINT32_MAX; else v = (w32)r;
        const w32 dot = dot_product_mod32(p, temp + t * lin, c->matrix, lin);
        const wide vq_wide = (wide)v * c->q_mod_p0;
        const w32 vq = barrett32_reduce_wide(p, (w32)vq_wide, (w32)(vq_wide >> 32));
        const w32 d = dot - vq;                                   /* reduce_sub (compact/primitive.rs) */
        crt_poly_out[t] = dot >= vq ? d : d + p->value;
    }
    free(temp);
    return ORC_OK;
}

/* ========================================================================== */
/* BigUintApproxSignedBasis<u32> — primus_decompose/src/big_integer/{basis,common}.rs */
/* ========================================================================== */
typedef struct { w32 mask; size_t index; uint32_t shr_bits, shl_bits; /* 0 = None */ } value_mask32_t;

struct orc_basis32 {
    size_t value_len, moduli_count, decompose_length;
    uint32_t log_basis, drop_bits;
    w32 basis, basis_minus_one, carry_mask;
    int mode; /* bit0 carry, bit1 adjust */
    w32 *threshold, *add;
    size_t carry_index; w32 carry_bit_mask;
    w32 *scalars, *scalars_residue, *modulus_sub_basis;
    value_mask32_t *masks;
};

static uint32_t clz32(w32 x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }
/* common.rs:83-103 */
static value_mask32_t value_mask32_new(w32 mask, uint32_t drop_bits) {
    value_mask32_t v; v.mask = mask; v.index = drop_bits / WBITS; v.shr_bits = drop_bits % WBITS;
    v.shl_bits = clz32(mask) < v.shr_bits ? WBITS - v.shr_bits : 0;
    return v;
}
/* common.rs:107-124 */
static value_mask32_t value_mask32_next(value_mask32_t v, uint32_t advance) {
    uint32_t shr = advance + v.shr_bits;
    if (shr >= WBITS) { v.index += 1; shr -= WBITS; }
    v.shr_bits = shr;
    v.shl_bits = clz32(v.mask) < shr ? WBITS - shr : 0;
    return v;
}
/* common.rs:132-140 */
static w32 value_mask32_get(const value_mask32_t *v, const w32 *value) {
    const w32 lower = value[v->index] >> v->shr_bits;
    if (v->shl_bits) return (lower | (value[v->index + 1] << v->shl_bits)) & v->mask;
    return lower & v->mask;
}

/* basis.rs:40-211 */
int orc_basis32_new(const orc_rns32 *rns, uint32_t log_basis, size_t reverse_length, orc_basis32 **out) {
    const size_t len = rns->value_len;
    const w32 *modulus = rns->product;
    if (modulus[len - 1] == 0 || log_basis == 0 || log_basis >= WBITS) return ORC_ERR_BAD_ARG; /* :50-51 */
    const uint32_t unused_bits = clz32(modulus[len - 1]);
    const w32 basis = (w32)1 << log_basis, bm1 = basis - 1;
    const uint32_t bits = WBITS * (uint32_t)len - unused_bits;
    size_t dlen = bits / log_basis;
    uint32_t drop = bits - (uint32_t)dlen * log_basis;
    if (reverse_length) { /* :63-68 */
        if (dlen < reverse_length) return ORC_ERR_BAD_ARG;
        dlen = reverse_length;
        drop = bits - (uint32_t)reverse_length * log_basis;
    }
    if (dlen == 0) return ORC_ERR_BAD_ARG;
    orc_basis32 *b = (orc_basis32 *)calloc(1, sizeof(*b));
    b->value_len = len; b->moduli_count = rns->count; b->decompose_length = dlen;
    b->log_basis = log_basis; b->drop_bits = drop; b->basis = basis; b->basis_minus_one = bm1;
    const int has_carry = drop > 0;
    if (has_carry) { const uint32_t cb = drop - 1; b->carry_index = cb / WBITS; b->carry_bit_mask = (w32)1 << (cb % WBITS); } /* :72-79 */
    b->carry_mask = log_basis == 1 ? ((w32)1 << 1) : (((w32)1 << log_basis) | ((w32)1 << (log_basis - 1))); /* :81-85 */
    /* split value :87-131 */
    w32 *split = (w32 *)calloc(len, sizeof(w32));
    w32 one[MAX_LIMBS] = {1};
    int has_split = 0;
    if (log_basis == 1) {
        if (drop != 0) {
            for (size_t i = 0; i < dlen; ++i) { big32_shl_assign(split, len, 1); split[0] |= 1; }
            big32_shl_assign(split, len, 1); split[0] |= 1;
            big32_shl_assign(split, len, drop - 1);
            has_split = big32_cmp(split, modulus, len) < 0;
        }
    } else {
        for (size_t i = 0; i < dlen; ++i) { big32_shl_assign(split, len, log_basis); split[0] |= bm1 >> 1; }
        if (drop > 0) { big32_shl_assign(split, len, 1); split[0] |= 1; big32_shl_assign(split, len, drop - 1); }
        else big32_add_assign(split, one, len);
        has_split = big32_cmp(split, modulus, len) < 0;
    }
    b->threshold = split;
    b->add = (w32 *)calloc(len, sizeof(w32));
    if (has_split) { /* make_adjust_add :137-147 */
        for (size_t i = 0; i < len; ++i) b->add[i] = ~(w32)0;
        b->add[len - 1] >>= unused_bits;
        w32 *qm1 = (w32 *)malloc(len * sizeof(w32));
        memcpy(qm1, modulus, len * sizeof(w32));
        big32_sub_assign(qm1, one, len);
        big32_sub_assign(b->add, qm1, len);
        free(qm1);
    } else {
        memset(split, 0, len * sizeof(w32));
    }
    b->mode = (has_split ? 2 : 0) | (has_carry ? 1 : 0);
    /* scalars :149-163 */
    b->scalars = (w32 *)calloc(dlen * len, sizeof(w32));
    for (size_t j = 0; j < dlen; ++j) {
        w32 *s = b->scalars + j * len;
        if (j == 0) { s[0] = 1; big32_shl_assign(s, len, drop); }
        else { memcpy(s, s - len, len * sizeof(w32)); big32_shl_assign(s, len, log_basis); }
    }
    b->scalars_residue = (w32 *)calloc(dlen * rns->count, sizeof(w32)); /* :165-173 */
    for (size_t j = 0; j < dlen; ++j) orc_rns32_decompose_to(rns, b->scalars + j * len, b->scalars_residue + j * rns->count);
    b->modulus_sub_basis = (w32 *)calloc(len, sizeof(w32)); /* :133-134 */
    {
        w32 bv[MAX_LIMBS] = {0};
        bv[0] = basis;
        memcpy(b->modulus_sub_basis, modulus, len * sizeof(w32));
        (void)big32_sub_assign(b->modulus_sub_basis, bv, len);
    }
    b->masks = (value_mask32_t *)malloc(dlen * sizeof(value_mask32_t)); /* :175-181 */
    b->masks[0] = value_mask32_new(bm1, drop);
    for (size_t j = 1; j < dlen; ++j) b->masks[j] = value_mask32_next(b->masks[j - 1], log_basis);
    *out = b;
    return ORC_OK;
}
void orc_basis32_free(orc_basis32 *b) {
    if (!b) return;
    free(b->threshold); free(b->add); free(b->scalars); free(b->scalars_residue); free(b->masks); free(b->modulus_sub_basis); free(b);
}
size_t orc_basis32_decompose_length(const orc_basis32 *b) { return b->decompose_length; }
uint32_t orc_basis32_log_basis(const orc_basis32 *b) { return b->log_basis; }
uint32_t orc_basis32_drop_bits(const orc_basis32 *b) { return b->drop_bits; }
uint32_t orc_basis32_basis_value(const orc_basis32 *b) { return b->basis; }
int orc_basis32_init_mode(const orc_basis32 *b) { return b->mode; }
const uint32_t *orc_basis32_scalars(const orc_basis32 *b) { return b->scalars; }
const uint32_t *orc_basis32_scalars_residue(const orc_basis32 *b) { return b->scalars_residue; }

/* basis.rs:326-367 */
void orc_basis32_init_value_carry_slice_inplace(const orc_basis32 *b, uint32_t *values, uint8_t *carries, size_t count) {
    const size_t len = b->value_len;
    for (size_t c = 0; c < count; ++c) {
        w32 *v = values + c * len;
        if ((b->mode & 2) && big32_cmp(v, b->threshold, len) >= 0) (void)big32_add_assign(v, b->add, len);
        carries[c] = (b->mode & 1) ? (uint8_t)((v[b->carry_index] & b->carry_bit_mask) != 0) : 0;
    }
}
/* basis.rs:371-420 */
void orc_basis32_init_value_carry_slice_to(const orc_basis32 *b, const uint32_t *values, uint32_t *adjusted, uint8_t *carries,
                                           size_t count) {
    memcpy(adjusted, values, count * b->value_len * sizeof(w32));
    orc_basis32_init_value_carry_slice_inplace(b, adjusted, carries, count);
}
/* common.rs:275-285 over a slice (:309-325) */
void orc_basis32_unsigned_decompose_slice_to(const orc_basis32 *b, size_t level, const uint32_t *values, uint32_t *digits,
                                             uint8_t *carries, size_t count) {
    const value_mask32_t *vm = &b->masks[level];
    for (size_t c = 0; c < count; ++c) {
        const w32 temp = value_mask32_get(vm, values + c * b->value_len) + (w32)carries[c];
        carries[c] = (uint8_t)((temp & b->carry_mask) != 0);
        digits[c] = temp & b->basis_minus_one;
    }
}
/* common.rs:255-272 over a slice (:289-306) */
void orc_basis32_decompose_slice_to(const orc_basis32 *b, size_t level, const uint32_t *values, uint32_t *decomposed,
                                    uint8_t *carries, size_t count) {
    const value_mask32_t *vm = &b->masks[level];
    const size_t len = b->value_len;
    for (size_t c = 0; c < count; ++c) {
        const w32 temp = value_mask32_get(vm, values + c * len) + (w32)carries[c];
        w32 *d = decomposed + c * len;
        carries[c] = (uint8_t)((temp & b->carry_mask) != 0);
        memset(d, 0, len * sizeof(w32));
        if (carries[c]) {
            if (temp <= b->basis_minus_one) {
                w32 tv[MAX_LIMBS] = {0};
                tv[0] = temp;
                memcpy(d, b->modulus_sub_basis, len * sizeof(w32));
                (void)big32_add_assign(d, tv, len);
            }
        } else {
            d[0] = temp;
        }
    }
}

/* ========================================================================== */
/* external product over U32DcrtTable — primus_lattice/src/glwe/{dcrt.rs,crt.rs}  */
/* ========================================================================== */
static void dcrt32_transform(const orc_u32_ntt *const *tables, size_t L, size_t n, uint32_t *poly) {
    for (size_t i = 0; i < L; ++i) orc_u32_ntt_transform_slice(tables[i], poly + i * n);
}
/* shared body of glwe/dcrt.rs:178-255 and :258-338 */
static void glev_row32(const orc_u32_ntt *const *tables, const orc_rns32 *rns, const orc_basis32 *basis, size_t k,
                       uint32_t *acc, const uint32_t *dcrt_glev, uint32_t *adjust, uint8_t *carries) {
    const size_t n = orc_u32_ntt_n(tables[0]), L = rns->count, W = L * n, glwe_len = (k + 1) * W;
    uint32_t *digits = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *multi = (uint32_t *)malloc(W * sizeof(uint32_t));
    for (size_t j = 0; j < basis->decompose_length; ++j) {
        const uint32_t *glwe = dcrt_glev + j * glwe_len;
        orc_basis32_unsigned_decompose_slice_to(basis, j, adjust, digits, carries, n);
        orc_rns32_wrapping_decompose_small_values_to(rns, digits, multi, n, basis->basis);
        dcrt32_transform(tables, L, n, multi);
        for (size_t c = 0; c <= k; ++c) /* add_dcrt_glwe_mul_dcrt_polynomial_assign, glwe/dcrt.rs:108-126 */
            for (size_t i = 0; i < L; ++i)
                orc_u32_reduce_add_mul_slice_assign(rns->moduli[i], acc + c * W + i * n, glwe + c * W + i * n, multi + i * n, n);
    }
    free(digits); free(multi);
}
/* glwe/dcrt.rs:178-255 */
void orc_add_dcrt32_glev_mul_crt_poly_assign(const orc_u32_ntt *const *tables, const orc_rns32 *rns, const orc_basis32 *basis,
                                             size_t k, uint32_t *acc, const uint32_t *dcrt_glev, const uint32_t *crt_poly) {
    const size_t n = orc_u32_ntt_n(tables[0]), len = rns->value_len;
    uint32_t *adjust = (uint32_t *)malloc(n * len * sizeof(uint32_t));
    uint8_t *carries = (uint8_t *)malloc(n);
    orc_rns32_compose_multiple_values_to(rns, crt_poly, adjust, n);      /* :219-224 */
    orc_basis32_init_value_carry_slice_inplace(basis, adjust, carries, n); /* :226 */
    glev_row32(tables, rns, basis, k, acc, dcrt_glev, adjust, carries);
    free(adjust); free(carries);
}
/* glwe/dcrt.rs:258-338 */
void orc_add_dcrt32_glev_mul_big_uint_poly_assign(const orc_u32_ntt *const *tables, const orc_rns32 *rns,
                                                  const orc_basis32 *basis, size_t k, uint32_t *acc, const uint32_t *dcrt_glev,
                                                  const uint32_t *big_uint_poly) {
    const size_t n = orc_u32_ntt_n(tables[0]), len = rns->value_len;
    uint32_t *adjust = (uint32_t *)malloc(n * len * sizeof(uint32_t));
    uint8_t *carries = (uint8_t *)malloc(n);
    orc_basis32_init_value_carry_slice_to(basis, big_uint_poly, adjust, carries, n); /* :301-306 */
    glev_row32(tables, rns, basis, k, acc, dcrt_glev, adjust, carries);
    free(adjust); free(carries);
}
/* glwe/crt.rs:200-227 */
void orc_mul_dcrt32_ggsw_to(const orc_u32_ntt *const *tables, const orc_rns32 *rns, const orc_basis32 *basis, size_t k,
                            const uint32_t *crt_glwe, const uint32_t *dcrt_ggsw, uint32_t *result) {
    const size_t W = rns->count * orc_u32_ntt_n(tables[0]);
    const size_t glwe_len = (k + 1) * W, glev_len = basis->decompose_length * glwe_len;
    memset(result, 0, glwe_len * sizeof(uint32_t)); /* :217 */
    for (size_t i = 0; i <= k; ++i)
        orc_add_dcrt32_glev_mul_crt_poly_assign(tables, rns, basis, k, result, dcrt_ggsw + i * glev_len, crt_glwe + i * W);
}
