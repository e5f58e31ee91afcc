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
    printf("oracle sanitize run ok\n");
    return 0;
}
