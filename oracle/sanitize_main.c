/* sanitize_main.c — drives every oracle entry point at small sizes; built with
 * -fsanitize=address,undefined by tests/test_oracle_sanitize.py (CPU only, test infrastructure). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pfhe_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
    const uint64_t Q61[3] = {2305843009211596801ull, 2305843009210023937ull, 2305843009208713217ull};
    /* U64NttTable round trips at several sizes, incl. the AVX-512 backend when present */
    for (uint32_t log_n = 0; log_n <= 11; ++log_n) {
        orc_u64_ntt *t = NULL;
        CHECK(orc_u64_ntt_new(log_n, Q61[0], &t) == ORC_OK);
        const size_t n = (size_t)1 << log_n;
        uint64_t *a = malloc(n * 8), *b = malloc(n * 8), *c = malloc(n * 8);
        for (size_t i = 0; i < n; ++i) a[i] = b[i] = c[i] = next_u64() % Q61[0];
        orc_u64_ntt_transform_slice(t, a);
        if (n >= 16 && orc_avx512_available()) {
            CHECK(orc_u64_ntt_forward_avx512(t, c, 0) == ORC_OK);
            CHECK(memcmp(a, c, n * 8) == 0);
        }
        orc_u64_ntt_inverse_transform_slice(t, a);
        CHECK(memcmp(a, b, n * 8) == 0);
        orc_u64_ntt_lazy_transform_slice(t, a);
        orc_u64_ntt_transform_monomial(t, 5, n / 2, a);
        orc_u64_ntt_transform_coeff_one_monomial(t, 1 % n, a);
        orc_u64_ntt_transform_coeff_minus_one_monomial(t, n - 1, a);
        free(a); free(b); free(c);
        orc_u64_ntt_free(t);
    }
    /* U32NttTable */
    for (uint32_t log_n = 0; log_n <= 10; ++log_n) {
        orc_u32_ntt *t = NULL;
        CHECK(orc_u32_ntt_new(log_n, 132120577u, &t) == ORC_OK);
        const size_t n = (size_t)1 << log_n;
        uint32_t *a = malloc(n * 4), *b = malloc(n * 4);
        for (size_t i = 0; i < n; ++i) a[i] = b[i] = (uint32_t)(next_u64() % 132120577u);
        orc_u32_ntt_transform_slice(t, a);
        orc_u32_ntt_inverse_transform_slice(t, a);
        CHECK(memcmp(a, b, n * 4) == 0);
        orc_u32_ntt_lazy_transform_slice(t, a);
        orc_u32_ntt_transform_monomial(t, 7, n / 2, a);
        free(a); free(b);
        orc_u32_ntt_free(t);
    }
    /* RNS base, gadget basis, external product, base conversion at N = 16 */
    {
        const uint32_t log_n = 4;
        const size_t n = 16, k = 1;
        orc_dcrt *d = NULL; orc_rns *r = NULL; orc_basis *bs = NULL;
        CHECK(orc_dcrt_new(log_n, Q61, 3, &d) == ORC_OK);
        CHECK(orc_rns_new(Q61, 3, &r) == ORC_OK);
        CHECK(orc_basis_new(r, 30, 0, &bs) == ORC_OK);
        const size_t ell = orc_basis_decompose_length(bs);
        const size_t W = 3 * n, glwe_len = (k + 1) * W, ggsw_len = (k + 1) * ell * glwe_len;
        uint64_t *glwe = malloc(glwe_len * 8), *ggsw = malloc(ggsw_len * 8), *out = malloc(glwe_len * 8);
        for (size_t i = 0; i < glwe_len; ++i) glwe[i] = next_u64() % Q61[(i / n) % 3];
        for (size_t i = 0; i < ggsw_len; ++i) ggsw[i] = next_u64() % Q61[(i / n) % 3];
        orc_mul_dcrt_ggsw_to(d, r, bs, k, glwe, ggsw, out);
        for (size_t i = 0; i < glwe_len; ++i) CHECK(out[i] < Q61[(i / n) % 3]);
        /* compose / decompose round trip */
        const size_t vl = orc_rns_value_len(r);
        uint64_t *big = malloc(n * vl * 8), *res = malloc(W * 8);
        orc_rns_compose_multiple_values_to(r, glwe, big, n);
        orc_rns_decompose_big_uint_values_to(r, big, res, n);
        CHECK(memcmp(res, glwe, W * 8) == 0);
        /* base conversion */
        const uint64_t P2[2] = {1152921504606584833ull, 1125899906826241ull};
        orc_rns *o2 = NULL, *o1 = NULL; orc_conv *cv = NULL, *ce = NULL;
        CHECK(orc_rns_new(P2, 2, &o2) == ORC_OK && orc_rns_new(P2, 1, &o1) == ORC_OK);
        CHECK(orc_conv_new(r, o2, &cv) == ORC_OK && orc_conv_new(r, o1, &ce) == ORC_OK);
        uint64_t *cout = malloc(2 * n * 8), *scratch = malloc(W * 8);
        orc_conv_fast_convert_array(cv, glwe, cout, n, scratch);
        CHECK(orc_conv_exact_convert_array(ce, glwe, cout, n) == ORC_OK);
        CHECK(orc_conv_exact_convert_array(cv, glwe, cout, n) != ORC_OK);
        free(cout); free(scratch); free(big); free(res); free(glwe); free(ggsw); free(out);
        orc_conv_free(cv); orc_conv_free(ce); orc_rns_free(o2); orc_rns_free(o1);
        orc_basis_free(bs); orc_rns_free(r); orc_dcrt_free(d);
    }
    /* the <u32> instantiations (pfhe_oracle_rns32.c): RNS base, gadget basis, external product over U32DcrtTable and
     * base conversion at N = 16; a nine-modulus base for the composition and the digits */
    {
        const uint32_t Q30[3] = {1073479681u, 1071513601u, 1070727169u};
        const uint32_t W9[9] = {1073707009u, 1073698817u, 1073692673u, 1073682433u, 1073668097u, 1073655809u, 1073651713u,
                                1073643521u, 1073620993u};
        const uint32_t log_n = 4;
        const size_t n = 16, k = 1;
        orc_u32_ntt *tabs[3] = {NULL, NULL, NULL};
        for (int i = 0; i < 3; ++i) CHECK(orc_u32_ntt_new(log_n, Q30[i], &tabs[i]) == ORC_OK);
        orc_rns32 *r = NULL; orc_basis32 *bs = NULL;
        CHECK(orc_rns32_new(Q30, 3, &r) == ORC_OK);
        CHECK(orc_basis32_new(r, 15, 0, &bs) == ORC_OK);
        const size_t ell = orc_basis32_decompose_length(bs);
        const size_t W = 3 * n, glwe_len = (k + 1) * W, ggsw_len = (k + 1) * ell * glwe_len;
        uint32_t *glwe = malloc(glwe_len * 4), *ggsw = malloc(ggsw_len * 4), *out = malloc(glwe_len * 4);
        for (size_t i = 0; i < glwe_len; ++i) glwe[i] = (uint32_t)(next_u64() % Q30[(i / n) % 3]);
        for (size_t i = 0; i < ggsw_len; ++i) ggsw[i] = (uint32_t)(next_u64() % Q30[(i / n) % 3]);
        orc_mul_dcrt32_ggsw_to((const orc_u32_ntt *const *)tabs, r, bs, k, glwe, ggsw, out);
        for (size_t i = 0; i < glwe_len; ++i) CHECK(out[i] < Q30[(i / n) % 3]);
        memset(out, 0, glwe_len * 4);
        orc_add_dcrt32_glev_mul_crt_poly_assign((const orc_u32_ntt *const *)tabs, r, bs, k, out, ggsw, glwe);
        const size_t vl = orc_rns32_value_len(r);
        uint32_t *big = malloc(n * vl * 4), *res = malloc(W * 4);
        orc_rns32_compose_multiple_values_to(r, glwe, big, n);
        orc_add_dcrt32_glev_mul_big_uint_poly_assign((const orc_u32_ntt *const *)tabs, r, bs, k, out, ggsw, big);
        orc_rns32_decompose_big_uint_values_to(r, big, res, n);
        CHECK(memcmp(res, glwe, W * 4) == 0);
        /* nine moduli: compose, decompose, balanced digits of every level */
        orc_rns32 *w = NULL; orc_basis32 *wb = NULL;
        CHECK(orc_rns32_new(W9, 9, &w) == ORC_OK);
        CHECK(orc_basis32_new(w, 20, 0, &wb) == ORC_OK);
        const size_t wl = orc_rns32_value_len(w), well = orc_basis32_decompose_length(wb);
        uint32_t *wr = malloc(9 * n * 4), *wbig = malloc(n * wl * 4), *wback = malloc(9 * n * 4), *dig = malloc(n * 4);
        uint8_t *car = malloc(n);
        for (size_t i = 0; i < 9 * n; ++i) wr[i] = (uint32_t)(next_u64() % W9[i / n]);
        orc_rns32_compose_multiple_values_to(w, wr, wbig, n);
        orc_rns32_decompose_big_uint_values_to(w, wbig, wback, n);
        CHECK(memcmp(wr, wback, 9 * n * 4) == 0);
        orc_basis32_init_value_carry_slice_inplace(wb, wbig, car, n);
        for (size_t l = 0; l < well; ++l) orc_basis32_unsigned_decompose_slice_to(wb, l, wbig, dig, car, n);
        /* base conversion: three -> two moduli (fast), three -> one (exact), nine -> three (fast) */
        const uint32_t P27[2] = {134215681u, 134176769u};
        orc_rns32 *o2 = NULL, *o1 = NULL; orc_conv32 *cv = NULL, *ce = NULL, *cw = NULL;
        CHECK(orc_rns32_new(P27, 2, &o2) == ORC_OK && orc_rns32_new(P27, 1, &o1) == ORC_OK);
        CHECK(orc_conv32_new(r, o2, &cv) == ORC_OK && orc_conv32_new(r, o1, &ce) == ORC_OK && orc_conv32_new(w, r, &cw) == ORC_OK);
        uint32_t *cout = malloc(3 * n * 4), *scratch = malloc(9 * n * 4);
        orc_conv32_fast_convert_array(cv, glwe, cout, n, scratch);
        for (size_t i = 0; i < 2 * n; ++i) CHECK(cout[i] < P27[i / n]);
        CHECK(orc_conv32_exact_convert_array(ce, glwe, cout, n) == ORC_OK);
        CHECK(orc_conv32_exact_convert_array(cv, glwe, cout, n) != ORC_OK);
        orc_conv32_fast_convert_array(cw, wr, cout, n, scratch);
        for (size_t i = 0; i < 3 * n; ++i) CHECK(cout[i] < Q30[i / n]);
        free(cout); free(scratch); free(wr); free(wbig); free(wback); free(dig); free(car);
        free(big); free(res); free(glwe); free(ggsw); free(out);
        orc_conv32_free(cv); orc_conv32_free(ce); orc_conv32_free(cw); orc_rns32_free(o2); orc_rns32_free(o1);
        orc_basis32_free(wb); orc_rns32_free(w); orc_basis32_free(bs); orc_rns32_free(r);
        for (int i = 0; i < 3; ++i) orc_u32_ntt_free(tabs[i]);
    }
    printf("oracle sanitize run ok\n");
    return 0;
}
