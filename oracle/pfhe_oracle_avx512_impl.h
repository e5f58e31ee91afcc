/*
 * pfhe_oracle_avx512_impl.h — body of the AVX-512 restatement, included twice by pfhe_oracle_avx512.c:
 *   SHIFTV = 64: BIT_SHIFT = 64 (AVX-512 DQ, approximate 64-bit quotients, any q < 2^62)
 *   SHIFTV = 52: BIT_SHIFT = 52 (AVX-512 IFMA, vpmadd52 products, q < 2^50: internal.rs:12,24,28; table.rs:166-186,236-256)
 *   SHIFTV = 32: BIT_SHIFT = 32 (AVX-512 DQ, 64-bit products of 32-bit operands, q < 2^30: internal.rs:16; table.rs:188-200)
 * FN(name) appends the variant's suffix, TGT carries its target attribute, AVAILABLE() its CPU check.
 * TEST / BENCH INFRASTRUCTURE ONLY (see pfhe_oracle.h).
 */
TGT static inline __m512i FN(small_mod)(__m512i x, __m512i m) { return _mm512_min_epu64(x, _mm512_sub_epi64(x, m)); }

/* utils/arithmetic.rs:94-127: high 64 bits of x*y without the lo*lo partial product (error <= 1) */
TGT static inline __m512i FN(mulhi_approx)(__m512i x, __m512i y) {
    const __m512i lo_mask = _mm512_set1_epi64(0xFFFFFFFFll);
    const __m512i x_hi = _mm512_shuffle_epi32(x, (_MM_PERM_ENUM)0xB1), y_hi = _mm512_shuffle_epi32(y, (_MM_PERM_ENUM)0xB1);
    const __m512i z_lo_hi = _mm512_mul_epu32(x, y_hi), z_hi_lo = _mm512_mul_epu32(x_hi, y), z_hi_hi = _mm512_mul_epu32(x_hi, y_hi);
    const __m512i sum_lo = _mm512_and_si512(z_lo_hi, lo_mask), sum_mid = _mm512_srli_epi64(z_lo_hi, 32);
    const __m512i sum_mid2 = _mm512_add_epi64(z_hi_lo, sum_lo);
    return _mm512_add_epi64(_mm512_add_epi64(z_hi_hi, sum_mid), _mm512_srli_epi64(sum_mid2, 32));
}

/* butterfly.rs:10-57, BIT_SHIFT = 64 */
TGT static inline void FN(fwd_bfly)(__m512i *x, __m512i *y, __m512i w, __m512i wp, __m512i neg_q, __m512i two_q) {
    *x = FN(small_mod)(*x, two_q);
#if SHIFTV == 52 /* butterfly.rs:30-35: exact 52-bit quotient, T in [0,2q) without a correction */
    const __m512i z = _mm512_setzero_si512();
    const __m512i qh = _mm512_madd52hi_epu64(z, wp, *y);
    const __m512i t = _mm512_and_si512(_mm512_madd52lo_epu64(_mm512_madd52lo_epu64(z, w, *y), qh, neg_q),
                                       _mm512_set1_epi64((1ll << 52) - 1));
#elif SHIFTV == 32 /* butterfly.rs:23-29: q < 2^30, y < 4q < 2^32: the quotient is the high half of a 64-bit product */
    const __m512i qh = _mm512_srli_epi64(_mm512_mullo_epi64(wp, *y), 32);
    const __m512i t = _mm512_add_epi64(_mm512_mullo_epi64(w, *y), _mm512_mullo_epi64(qh, neg_q));
#else
    const __m512i qh = FN(mulhi_approx)(wp, *y);
    __m512i t = _mm512_add_epi64(_mm512_mullo_epi64(w, *y), _mm512_mullo_epi64(qh, neg_q)); /* [0,4q) */
    t = FN(small_mod)(t, two_q);
#endif
    *y = _mm512_add_epi64(*x, _mm512_sub_epi64(two_q, t));
    *x = _mm512_add_epi64(*x, t);
}

/* one stage with butterfly distance t >= 8 over `n` values whose first group uses roots[ri0] */
TGT static void FN(stage_t8)(uint64_t *v, size_t n, size_t t, const uint64_t *w, const uint64_t *wp, size_t ri0,
                         __m512i neg_q, __m512i two_q) {
    size_t ri = ri0;
    for (size_t c = 0; c < n; c += 2 * t, ++ri) {
        const __m512i vw = _mm512_set1_epi64((long long)w[ri]), vwp = _mm512_set1_epi64((long long)wp[ri]);
        for (size_t j = 0; j < t; j += 8) {
            __m512i x = _mm512_loadu_si512(v + c + j), y = _mm512_loadu_si512(v + c + j + t);
            FN(fwd_bfly)(&x, &y, vw, vwp, neg_q, two_q);
            _mm512_storeu_si512(v + c + j, x);
            _mm512_storeu_si512(v + c + j + t, y);
        }
    }
}

/* the last three stages (distances 4, 2, 1) on 16 consecutive values at a time; `m4` = number of
 * distance-4 groups in this sub-transform, whose roots start at roots[ri4] (then 2*ri4, 4*ri4) */
TGT static void FN(stages_t4_t2_t1)(uint64_t *v, size_t n, const uint64_t *w, const uint64_t *wp, size_t ri4,
                                __m512i neg_q, __m512i two_q, __m512i q, int canonical) {
    const __m512i ix4 = _mm512_setr_epi64(0, 1, 2, 3, 8, 9, 10, 11), iy4 = _mm512_setr_epi64(4, 5, 6, 7, 12, 13, 14, 15);
    const __m512i iw4 = _mm512_setr_epi64(0, 0, 0, 0, 1, 1, 1, 1);
    const __m512i ix2 = _mm512_setr_epi64(0, 1, 4, 5, 8, 9, 12, 13), iy2 = _mm512_setr_epi64(2, 3, 6, 7, 10, 11, 14, 15);
    const __m512i iw2 = _mm512_setr_epi64(0, 0, 1, 1, 2, 2, 3, 3);
    const __m512i oa2 = _mm512_setr_epi64(0, 1, 8, 9, 2, 3, 10, 11), ob2 = _mm512_setr_epi64(4, 5, 12, 13, 6, 7, 14, 15);
    const __m512i ix1 = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14), iy1 = _mm512_setr_epi64(1, 3, 5, 7, 9, 11, 13, 15);
    const __m512i oa1 = _mm512_setr_epi64(0, 8, 1, 9, 2, 10, 3, 11), ob1 = _mm512_setr_epi64(4, 12, 5, 13, 6, 14, 7, 15);
    size_t r4 = ri4, r2 = 2 * ri4, r1 = 4 * ri4;
    for (size_t c = 0; c < n; c += 16, r4 += 2, r2 += 4, r1 += 8) {
        __m512i a = _mm512_loadu_si512(v + c), b = _mm512_loadu_si512(v + c + 8);
        /* distance 4 */
        __m512i x = _mm512_permutex2var_epi64(a, ix4, b), y = _mm512_permutex2var_epi64(a, iy4, b);
        __m512i vw = _mm512_permutexvar_epi64(iw4, _mm512_maskz_loadu_epi64(0x03, w + r4));
        __m512i vp = _mm512_permutexvar_epi64(iw4, _mm512_maskz_loadu_epi64(0x03, wp + r4));
        FN(fwd_bfly)(&x, &y, vw, vp, neg_q, two_q);
        a = _mm512_permutex2var_epi64(x, ix4, y); /* [X0-3, Y0-3] */
        b = _mm512_permutex2var_epi64(x, iy4, y); /* [X4-7, Y4-7] */
        /* distance 2 */
        x = _mm512_permutex2var_epi64(a, ix2, b);
        y = _mm512_permutex2var_epi64(a, iy2, b);
        vw = _mm512_permutexvar_epi64(iw2, _mm512_maskz_loadu_epi64(0x0F, w + r2));
        vp = _mm512_permutexvar_epi64(iw2, _mm512_maskz_loadu_epi64(0x0F, wp + r2));
        FN(fwd_bfly)(&x, &y, vw, vp, neg_q, two_q);
        a = _mm512_permutex2var_epi64(x, oa2, y);
        b = _mm512_permutex2var_epi64(x, ob2, y);
        /* distance 1 */
        x = _mm512_permutex2var_epi64(a, ix1, b);
        y = _mm512_permutex2var_epi64(a, iy1, b);
        vw = _mm512_loadu_si512(w + r1);
        vp = _mm512_loadu_si512(wp + r1);
        FN(fwd_bfly)(&x, &y, vw, vp, neg_q, two_q);
        if (canonical) { /* [0,4q) -> [0,q) */
            x = FN(small_mod)(FN(small_mod)(x, two_q), q);
            y = FN(small_mod)(FN(small_mod)(y, two_q), q);
        }
        _mm512_storeu_si512(v + c, _mm512_permutex2var_epi64(x, oa1, y));
        _mm512_storeu_si512(v + c + 8, _mm512_permutex2var_epi64(x, ob1, y));
    }
}

/* transform.rs:13-260: sub-transform of `n` values whose first stage (distance n/2) has its single
 * group at roots[ri]; the following stage's groups start at roots[2*ri], and so on */
TGT static void FN(forward_rec)(uint64_t *v, size_t n, const uint64_t *w, const uint64_t *wp, size_t ri, __m512i neg_q,
                            __m512i two_q, __m512i q, int canonical) {
    if (n > 1024) { /* depth-first: one stage, then the two halves */
        FN(stage_t8)(v, n, n >> 1, w, wp, ri, neg_q, two_q);
        FN(forward_rec)(v, n >> 1, w, wp, 2 * ri, neg_q, two_q, q, canonical);
        FN(forward_rec)(v + (n >> 1), n >> 1, w, wp, 2 * ri + 1, neg_q, two_q, q, canonical);
        return;
    }
    size_t t = n >> 1, r = ri;
    for (; t >= 8; t >>= 1, r <<= 1) FN(stage_t8)(v, n, t, w, wp, r, neg_q, two_q);
    FN(stages_t4_t2_t1)(v, n, w, wp, r, neg_q, two_q, q, canonical);
}

/* U64NttTable::transform_slice / lazy_transform_slice through the AVX-512 backend (n >= 16) */
TGT int FN(orc_u64_ntt_forward_avx512)(const orc_u64_ntt *t, uint64_t *values, int lazy) {
    const size_t n = orc_u64_ntt_n(t);
    if (n < 16 || !AVAILABLE()) return ORC_ERR_BAD_ARG;
    const uint64_t qv = orc_u64_ntt_modulus(t);
    if (SHIFTV == 52 && (qv >= (1ull << 50) || !orc_u64_ntt_roots_precon52(t))) return ORC_ERR_BAD_ARG; /* internal.rs:24 */
    if (SHIFTV == 32 && (qv >= (1ull << 30) || !orc_u64_ntt_roots_precon32(t))) return ORC_ERR_BAD_ARG; /* internal.rs:16 */
    const __m512i q = _mm512_set1_epi64((long long)qv), two_q = _mm512_set1_epi64((long long)(qv << 1));
    const __m512i neg_q = _mm512_set1_epi64(-(long long)qv);
    FN(forward_rec)(values, n, orc_u64_ntt_roots(t),
                    SHIFTV == 52 ? orc_u64_ntt_roots_precon52(t) : SHIFTV == 32 ? orc_u64_ntt_roots_precon32(t) : orc_u64_ntt_roots_precon64(t),
                    1, neg_q, two_q, q, !lazy);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * inverse transform — prime64/avx512/transform.rs:205-423 with BIT_SHIFT = 64, input_mod_factor = 1
 * (table.rs:257-272).  Twiddle of the stage with `m` groups over the whole transform: inv_roots[1 + N - 2m + g]
 * (the reference reaches the same entries by walking w_idx with its per-depth deltas).
 * ------------------------------------------------------------------------------------------------ */

/* utils/arithmetic.rs:19-60: exact high 64 bits of x*y */
TGT static inline __m512i FN(mulhi_exact)(__m512i x, __m512i y) {
    const __m512i lo_mask = _mm512_set1_epi64(0xFFFFFFFFll);
    const __m512i x_hi = _mm512_shuffle_epi32(x, (_MM_PERM_ENUM)0xB1), y_hi = _mm512_shuffle_epi32(y, (_MM_PERM_ENUM)0xB1);
    const __m512i z_lo_lo = _mm512_mul_epu32(x, y), z_lo_hi = _mm512_mul_epu32(x, y_hi);
    const __m512i z_hi_lo = _mm512_mul_epu32(x_hi, y), z_hi_hi = _mm512_mul_epu32(x_hi, y_hi);
    const __m512i sum_tmp = _mm512_add_epi64(z_lo_hi, _mm512_srli_epi64(z_lo_lo, 32));
    const __m512i sum_lo = _mm512_and_si512(sum_tmp, lo_mask), sum_mid = _mm512_srli_epi64(sum_tmp, 32);
    const __m512i sum_mid2 = _mm512_add_epi64(z_hi_lo, sum_lo);
    return _mm512_add_epi64(_mm512_add_epi64(z_hi_hi, sum_mid), _mm512_srli_epi64(sum_mid2, 32));
}

/* butterfly.rs:58-117, BIT_SHIFT = 64: X' = X + Y mod 2q, Y' = W * (X - Y + 2q) mod~ q in [0,2q) */
TGT static inline void FN(inv_bfly)(__m512i *x, __m512i *y, __m512i w, __m512i wp, __m512i neg_q, __m512i two_q,
                                int input_less_than_mod) {
    const __m512i y_minus_2q = _mm512_sub_epi64(*y, two_q);
    const __m512i t = _mm512_sub_epi64(*x, y_minus_2q);
    if (input_less_than_mod) {
        *x = _mm512_add_epi64(*x, *y);
    } else {
        *x = _mm512_add_epi64(*x, y_minus_2q);
        const __mmask8 neg = _mm512_movepi64_mask(*x);
        *x = _mm512_mask_add_epi64(*x, neg, *x, two_q);
    }
#if SHIFTV == 52 /* butterfly.rs:97-102 */
    const __m512i z = _mm512_setzero_si512();
    const __m512i qh = _mm512_madd52hi_epu64(z, wp, t);
    *y = _mm512_and_si512(_mm512_madd52lo_epu64(_mm512_madd52lo_epu64(z, qh, neg_q), w, t), _mm512_set1_epi64((1ll << 52) - 1));
#elif SHIFTV == 32 /* butterfly.rs:90-96 */
    const __m512i qh = _mm512_srli_epi64(_mm512_mullo_epi64(wp, t), 32);
    *y = _mm512_add_epi64(_mm512_mullo_epi64(qh, neg_q), _mm512_mullo_epi64(w, t));
#else
    const __m512i qh = FN(mulhi_approx)(wp, t);
    *y = FN(small_mod)(_mm512_add_epi64(_mm512_mullo_epi64(w, t), _mm512_mullo_epi64(qh, neg_q)), two_q);
#endif
}

/* stages at distances 1, 2, 4 on 16 consecutive values at a time (stages.rs: inv_t1, inv_t2, inv_t4).
 * r1 / r2 / r4: index of the first twiddle of this sub-transform's distance-1 / 2 / 4 stage. */
TGT static void FN(inv_stages_t1_t2_t4)(uint64_t *v, size_t n, const uint64_t *w, const uint64_t *wp, size_t r1, size_t r2,
                                    size_t r4, __m512i neg_q, __m512i two_q, int input_less_than_mod) {
    const __m512i even = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14), odd = _mm512_setr_epi64(1, 3, 5, 7, 9, 11, 13, 15);
    /* after distance 1: X[k] = element 2k, Y[k] = element 2k+1; distance-2 operands: elements {0,1,4,5,8,9,12,13} and
     * {2,3,6,7,10,11,14,15} */
    const __m512i x2i = _mm512_setr_epi64(0, 8, 2, 10, 4, 12, 6, 14), y2i = _mm512_setr_epi64(1, 9, 3, 11, 5, 13, 7, 15);
    const __m512i iw2 = _mm512_setr_epi64(0, 0, 1, 1, 2, 2, 3, 3);
    /* after distance 2: x lanes = elements {0,1,4,5,8,9,12,13}, y lanes = {2,3,6,7,10,11,14,15}; distance-4 operands:
     * elements {0,1,2,3,8,9,10,11} and {4,5,6,7,12,13,14,15} */
    const __m512i x4i = _mm512_setr_epi64(0, 1, 8, 9, 4, 5, 12, 13), y4i = _mm512_setr_epi64(2, 3, 10, 11, 6, 7, 14, 15);
    const __m512i iw4 = _mm512_setr_epi64(0, 0, 0, 0, 1, 1, 1, 1);
    /* after distance 4: x lanes = elements {0..3, 8..11}, y lanes = {4..7, 12..15} */
    const __m512i oa = _mm512_setr_epi64(0, 1, 2, 3, 8, 9, 10, 11), ob = _mm512_setr_epi64(4, 5, 6, 7, 12, 13, 14, 15);
    for (size_t c = 0; c < n; c += 16, r1 += 8, r2 += 4, r4 += 2) {
        const __m512i a = _mm512_loadu_si512(v + c), b = _mm512_loadu_si512(v + c + 8);
        __m512i x = _mm512_permutex2var_epi64(a, even, b), y = _mm512_permutex2var_epi64(a, odd, b);
        FN(inv_bfly)(&x, &y, _mm512_loadu_si512(w + r1), _mm512_loadu_si512(wp + r1), neg_q, two_q, input_less_than_mod);
        __m512i x2 = _mm512_permutex2var_epi64(x, x2i, y), y2 = _mm512_permutex2var_epi64(x, y2i, y);
        FN(inv_bfly)(&x2, &y2, _mm512_permutexvar_epi64(iw2, _mm512_maskz_loadu_epi64(0x0F, w + r2)),
                 _mm512_permutexvar_epi64(iw2, _mm512_maskz_loadu_epi64(0x0F, wp + r2)), neg_q, two_q, 0);
        __m512i x4 = _mm512_permutex2var_epi64(x2, x4i, y2), y4 = _mm512_permutex2var_epi64(x2, y4i, y2);
        FN(inv_bfly)(&x4, &y4, _mm512_permutexvar_epi64(iw4, _mm512_maskz_loadu_epi64(0x03, w + r4)),
                 _mm512_permutexvar_epi64(iw4, _mm512_maskz_loadu_epi64(0x03, wp + r4)), neg_q, two_q, 0);
        _mm512_storeu_si512(v + c, _mm512_permutex2var_epi64(x4, oa, y4));
        _mm512_storeu_si512(v + c + 8, _mm512_permutex2var_epi64(x4, ob, y4));
    }
}

/* one stage with butterfly distance t >= 8 (stages.rs: inv_t8); the first group uses inv_roots[ri0] */
TGT static void FN(inv_stage_t8)(uint64_t *v, size_t n, size_t t, const uint64_t *w, const uint64_t *wp, size_t ri0,
                             __m512i neg_q, __m512i two_q) {
    size_t ri = ri0;
    for (size_t c = 0; c < n; c += 2 * t, ++ri) {
        const __m512i vw = _mm512_set1_epi64((long long)w[ri]), vwp = _mm512_set1_epi64((long long)wp[ri]);
        for (size_t j = 0; j < t; j += 8) {
            __m512i x = _mm512_loadu_si512(v + c + j), y = _mm512_loadu_si512(v + c + j + t);
            FN(inv_bfly)(&x, &y, vw, vwp, neg_q, two_q, 0);
            _mm512_storeu_si512(v + c + j, x);
            _mm512_storeu_si512(v + c + j + t, y);
        }
    }
}

/* transform.rs:237-334: the sub-transform over the n values starting at element `e0` of a transform of big_n points
 * runs its stages at distances 1 .. n/4 and leaves its own last stage (distance n/2) to its caller: above 1024 points
 * the caller recurses into the two halves and then runs THEIR last stage (distance n/4 here, two groups) in one sweep;
 * the last stage of the whole transform is the fused loop of orc_u64_ntt_inverse_avx512.
 * Index of the first twiddle of the stage at distance t: 1 + N - N/t + e0/(2t). */
TGT static void FN(inverse_rec)(uint64_t *v, size_t n, size_t big_n, size_t e0, const uint64_t *w, const uint64_t *wp,
                            __m512i neg_q, __m512i two_q, int depth0) {
#define RI(t) (1 + big_n - big_n / (t) + e0 / (2 * (t)))
    if (n <= 1024) { /* breadth-first; inputs are below q only for the very first stage of an undivided transform */
        FN(inv_stages_t1_t2_t4)(v, n, w, wp, RI(1), RI(2), RI(4), neg_q, two_q, depth0);
        for (size_t t = 8; 4 * t <= n; t <<= 1) FN(inv_stage_t8)(v, n, t, w, wp, RI(t), neg_q, two_q);
        return;
    }
    FN(inverse_rec)(v, n >> 1, big_n, e0, w, wp, neg_q, two_q, 0);
    FN(inverse_rec)(v + (n >> 1), n >> 1, big_n, e0 + (n >> 1), w, wp, neg_q, two_q, 0);
    FN(inv_stage_t8)(v, n, n >> 2, w, wp, RI(n >> 2), neg_q, two_q);
#undef RI
}

/* U64NttTable::inverse_transform_slice / lazy_inverse_transform_slice through the AVX-512 backend (n >= 16) */
TGT int FN(orc_u64_ntt_inverse_avx512)(const orc_u64_ntt *t, uint64_t *values, int lazy) {
    const size_t n = orc_u64_ntt_n(t);
    if (n < 16 || !AVAILABLE()) return ORC_ERR_BAD_ARG;
    const uint64_t qv = orc_u64_ntt_modulus(t);
    if (SHIFTV == 52 && (qv >= (1ull << 50) || !orc_u64_ntt_inv_roots_precon52(t))) return ORC_ERR_BAD_ARG; /* internal.rs:28 */
    if (SHIFTV == 32 && (qv >= (1ull << 30) || !orc_u64_ntt_inv_roots_precon32(t))) return ORC_ERR_BAD_ARG; /* this restatement keeps the forward bound for both directions (the table holds 32-bit preconditioners for q < 2^30 only) */
    const __m512i q = _mm512_set1_epi64((long long)qv), two_q = _mm512_set1_epi64((long long)(qv << 1));
    const __m512i neg_q = _mm512_set1_epi64(-(long long)qv);
    FN(inverse_rec)(values, n, n, 0, orc_u64_ntt_inv_roots(t),
                    SHIFTV == 52 ? orc_u64_ntt_inv_roots_precon52(t) : SHIFTV == 32 ? orc_u64_ntt_inv_roots_precon32(t) : orc_u64_ntt_inv_roots_precon64(t),
                    neg_q, two_q, 1);
    /* transform.rs:336-421: final stage with N^-1 (x half) and N^-1 * w (y half), exact quotients */
    const uint64_t inv_n = orc_u64_ntt_inv_n(t), inv_n_w = orc_u64_ntt_inv_n_w(t);
    const __m512i v_inv_n = _mm512_set1_epi64((long long)inv_n), v_inv_n_w = _mm512_set1_epi64((long long)inv_n_w);
    /* MultiplyFactor::new(value, BIT_SHIFT, q).quotient() = floor(value * 2^BIT_SHIFT / q), transform.rs:344-348 */
    const __m512i v_inv_n_p = _mm512_set1_epi64((long long)(SHIFTV == 64 ? orc_shoup_quotient(inv_n, qv) : orc_multiply_factor_quotient(inv_n, SHIFTV, qv)));
    const __m512i v_inv_n_w_p = _mm512_set1_epi64((long long)(SHIFTV == 64 ? orc_shoup_quotient(inv_n_w, qv) : orc_multiply_factor_quotient(inv_n_w, SHIFTV, qv)));
    const size_t h = n >> 1;
    for (size_t j = 0; j < h; j += 8) {
        __m512i x = _mm512_loadu_si512(values + j), y = _mm512_loadu_si512(values + j + h);
        const __m512i y_minus_2q = _mm512_sub_epi64(y, two_q);
        const __m512i s = FN(small_mod)(_mm512_add_epi64(x, y), two_q);
        const __m512i d = _mm512_sub_epi64(x, y_minus_2q);
#if SHIFTV == 52 /* transform.rs:388-397 */
        const __m512i z = _mm512_setzero_si512(), m52 = _mm512_set1_epi64((1ll << 52) - 1);
        x = _mm512_and_si512(_mm512_madd52lo_epu64(_mm512_madd52lo_epu64(z, v_inv_n, s), _mm512_madd52hi_epu64(z, v_inv_n_p, s), neg_q), m52);
        y = _mm512_and_si512(_mm512_madd52lo_epu64(_mm512_madd52lo_epu64(z, v_inv_n_w, d), _mm512_madd52hi_epu64(z, v_inv_n_w_p, d), neg_q), m52);
#elif SHIFTV == 32 /* transform.rs:375-387 */
        x = _mm512_add_epi64(_mm512_mullo_epi64(v_inv_n, s), _mm512_mullo_epi64(_mm512_srli_epi64(_mm512_mullo_epi64(v_inv_n_p, s), 32), neg_q));
        y = _mm512_add_epi64(_mm512_mullo_epi64(v_inv_n_w, d), _mm512_mullo_epi64(_mm512_srli_epi64(_mm512_mullo_epi64(v_inv_n_w_p, d), 32), neg_q));
#else
        x = _mm512_add_epi64(_mm512_mullo_epi64(v_inv_n, s), _mm512_mullo_epi64(FN(mulhi_exact)(v_inv_n_p, s), neg_q));
        y = _mm512_add_epi64(_mm512_mullo_epi64(v_inv_n_w, d), _mm512_mullo_epi64(FN(mulhi_exact)(v_inv_n_w_p, d), neg_q));
#endif
        if (!lazy) {
            x = FN(small_mod)(x, q);
            y = FN(small_mod)(y, q);
        }
        _mm512_storeu_si512(values + j, x);
        _mm512_storeu_si512(values + j + h, y);
    }
    return ORC_OK;
}
