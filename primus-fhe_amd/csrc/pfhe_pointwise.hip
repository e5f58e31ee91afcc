// pfhe_pointwise.hip — streaming modular kernels: Barrett multiply / multiply-accumulate,
// monomial transforms, synthetic fill.  All HBM-bound: 16-byte accesses, grid-stride.
//
//   mul_assign      a[i] = a[i]*b[i] mod q_r        DcrtPolynomial::mul_assign, dcrt/mul.rs:176-187
//   add_mul_assign  acc[i] = a[i]*b[i] + acc[i]     DcrtPolynomial::add_mul_assign, dcrt/mod.rs:105-123
//   mul_to          out[i] = a[i]*b[i]              NttPolynomial::mul_to, ntt/mul.rs:100-107; dcrt/mul.rs:232-250
//   mul_add_to      out[i] = a[i]*b[i] + c[i]       NttPolynomial::mul_add_to, ntt/mod.rs:169-187
// (per limb: BarrettModulus::reduce_mul / reduce_mul_add, primus_modulus/src/barrett/ops.rs:276-315)
#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"
#include "pfhe_pointwise.hpp"

namespace pfhe {

namespace {

constexpr int kPwThreads = 256;
// Launch shape as in pfhe_elementwise.hip: one vector per thread, one workgroup per 256 vectors (no grid-stride
// below 2^31 workgroups), non-temporal streams: mul_assign with a per-element multiplicand 4.10 -> 3.24 ms on 6 GiB.
constexpr int kPwUnroll = 1;  // vectors per thread and iteration

using pw_vec = __attribute__((__vector_size__(2 * sizeof(u64)))) u64;
__device__ __forceinline__ pw_vec pw_load(const u64 *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const pw_vec *>(p));
}
__device__ __forceinline__ void pw_store(u64 *p, pw_vec v) {
    __builtin_nontemporal_store(v, reinterpret_cast<pw_vec *>(p));
}

struct Bar {
    u64 q, lo, hi;
};

__device__ __forceinline__ Bar load_bar(const NttPrime *__restrict__ primes, u32 limb) {
    const NttPrime *P = primes + limb;
    return Bar{P->q, P->bar_lo, P->bar_hi};
}

// out = a*b (HAS_C: + c), element by element; `out` may alias `a` and/or `c`, which gives the
// in-place forms (mul_assign: out = a; add_mul_assign: out = c).  Each thread handles UNROLL
// 16-byte vectors per iteration, one workgroup-stride apart, so that several independent loads
// are in flight.
// PM: every modulus is pseudo-Mersenne (q = 2^K - c): the folding multiply of PmArith replaces the 128-bit Barrett
// reduction (a third of the instructions); both return the canonical residue.
template <bool HAS_C, bool PAIR, bool PM>
// (no __restrict__ on the data pointers: out may be a and/or c, and b may be a or out — mul_assign(a, a) squares in
// place; every word is read before the same thread writes it)
__global__ __launch_bounds__(kPwThreads) void pointwise_kernel(u64 *out, const u64 *a, const u64 *b,
                                                               const u64 *c, const NttPrime *__restrict__ primes,
                                                               u32 L, u32 log_n, u64 len, u64 len_b, u64 group_words) {
    constexpr u64 V = PAIR ? 2 : 1;
    constexpr int UNROLL = kPwUnroll;
    const u64 nvec = len / V;
    const bool shared_b = len_b != len;
    const u32 group_units = group_words ? (u32)(group_words >> log_n) / L : 1;
    const u64 tile = (u64)gridDim.x * blockDim.x;
    for (u64 v0 = (u64)blockIdx.x * blockDim.x + threadIdx.x; v0 < nvec; v0 += tile * UNROLL) {
        u64 av[UNROLL][2], bv[UNROLL][2], cv[UNROLL][2];
        Bar m[UNROLL];
        u32 limb_of[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 v = v0 + tile * u;
            if (v >= nvec) continue;
            const u64 i = v * V;
            // limb-polynomial index: the same for a whole wave once a polynomial spans a wave's 128 words, which
            // moves the divisions below to the scalar unit
            u32 p = (u32)(i >> log_n);
            if (log_n >= 7) p = __builtin_amdgcn_readfirstlane(p);
            const u32 unit_idx = p / L, limb = p - unit_idx * L;
            limb_of[u] = limb;
            if constexpr (!PM) m[u] = load_bar(primes, limb);
            u64 ib = i;
            if (shared_b || group_words) {
                // shared: the one unit of b; grouped: one unit per `group_units` consecutive units of a
                const u64 unit_b = group_words ? unit_idx / group_units : 0;
                ib = ((unit_b * L + limb) << log_n) + (i & (((u64)1 << log_n) - 1));
            }
            if constexpr (PAIR) {
                const pw_vec x = pw_load(a + i);
                const pw_vec y = shared_b ? *reinterpret_cast<const pw_vec *>(b + ib) : pw_load(b + ib);
                av[u][0] = x[0]; av[u][1] = x[1]; bv[u][0] = y[0]; bv[u][1] = y[1];
                if constexpr (HAS_C) {
                    const pw_vec z = pw_load(c + i);
                    cv[u][0] = z[0]; cv[u][1] = z[1];
                }
            } else {
                av[u][0] = a[i];
                bv[u][0] = b[ib];
                if constexpr (HAS_C) cv[u][0] = c[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 v = v0 + tile * u;
            if (v >= nvec) continue;
            const u64 i = v * V;
            u64 r[2];
            if constexpr (PM) {
                const PmArith ar(primes + limb_of[u]);
#pragma unroll
                for (int e = 0; e < (int)V; ++e) {
                    const u64 prod = ar.mul_any(av[u][e], bv[u][e]);  // [0, 2q)
                    if constexpr (!HAS_C) r[e] = ar.reduce_2q(prod);
                    else r[e] = ar.reduce_4q(prod + cv[u][e]);      // < 3q
                }
            } else {
#pragma unroll
                for (int e = 0; e < (int)V; ++e) {
                    if constexpr (!HAS_C) r[e] = mul_mod_barrett(av[u][e], bv[u][e], m[u].q, m[u].lo, m[u].hi);
                    else r[e] = mul_add_mod_barrett(av[u][e], bv[u][e], cv[u][e], m[u].q, m[u].lo, m[u].hi);
                }
            }
            if constexpr (PAIR) pw_store(out + i, pw_vec{r[0], r[1]});
            else out[i] = r[0];
        }
    }
}

// GLWE butterfly (a, b) = (a + s, (a - s) * w), canonical in and out:
//   FACTOR: w = ShoupFactor pairs (value, quotient)   DcrtPolynomial::butterfly_mul_factor_to, dcrt/mul.rs:15-30,196-222
//   else  : w = plain residues, Barrett product        DcrtPolynomial::butterfly_mul_to, dcrt/mod.rs:125-160
template <bool FACTOR, bool PAIR, bool PM>
__global__ __launch_bounds__(kPwThreads) void butterfly_kernel(u64 *__restrict__ a, const u64 *__restrict__ s,
                                                               const u64 *__restrict__ w, u64 *__restrict__ b,
                                                               const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                               u64 len, bool shared_w) {
    constexpr u64 V = PAIR ? 2 : 1;
    const u64 nvec = len / V;
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (u64)gridDim.x * blockDim.x) {
        const u64 i = v * V;
        u32 p = (u32)(i >> log_n);  // wave-uniform once a polynomial spans a wave's 128 words
        if (log_n >= 7) p = __builtin_amdgcn_readfirstlane(p);
        const u32 limb = p % L;
        const NttPrime *P = primes + limb;
        const u64 q = P->q;
        const u64 iw = shared_w ? (((u64)limb << log_n) + (i & (((u64)1 << log_n) - 1))) : i;
        u64 x[2], y[2], wv[2], wq[2];
        if constexpr (PAIR) {
            const pw_vec xv = pw_load(a + i), yv = pw_load(s + i);
            x[0] = xv[0]; x[1] = xv[1]; y[0] = yv[0]; y[1] = yv[1];
            if constexpr (FACTOR) {
                const pw_vec f0 = shared_w ? *reinterpret_cast<const pw_vec *>(w + 2 * iw) : pw_load(w + 2 * iw);
                const pw_vec f1 = shared_w ? *reinterpret_cast<const pw_vec *>(w + 2 * iw + 2) : pw_load(w + 2 * iw + 2);
                wv[0] = f0[0]; wq[0] = f0[1]; wv[1] = f1[0]; wq[1] = f1[1];
            } else {
                const pw_vec f = shared_w ? *reinterpret_cast<const pw_vec *>(w + iw) : pw_load(w + iw);
                wv[0] = f[0]; wv[1] = f[1];
            }
        } else {
            x[0] = a[i]; y[0] = s[i];
            if constexpr (FACTOR) { wv[0] = w[2 * iw]; wq[0] = w[2 * iw + 1]; } else wv[0] = w[iw];
        }
        u64 ra[2], rb[2];
#pragma unroll
        for (int e = 0; e < (int)V; ++e) {
            ra[e] = add_mod(x[e], y[e], q);
            const u64 d = sub_mod(x[e], y[e], q);
            if constexpr (FACTOR) {
                rb[e] = mul_shoup(d, wv[e], wq[e], q);
            } else if constexpr (PM) {
                const PmArith ar(P);
                rb[e] = ar.reduce_2q(ar.mul_any(d, wv[e]));
            } else {
                rb[e] = mul_mod_barrett(d, wv[e], q, P->bar_lo, P->bar_hi);
            }
        }
        if constexpr (PAIR) {
            pw_store(a + i, pw_vec{ra[0], ra[1]});
            pw_store(b + i, pw_vec{rb[0], rb[1]});
        } else {
            a[i] = ra[0];
            b[i] = rb[0];
        }
    }
}

__device__ __forceinline__ u64 splitmix64(u64 seed, u64 i) {
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kPwThreads) void fill_uniform_kernel(u64 *__restrict__ dst, u64 len,
                                                                  const u64 *__restrict__ moduli, u64 count,
                                                                  u64 poly_len, u64 seed) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (u64)gridDim.x * blockDim.x) {
        const u64 q = moduli[(i / poly_len) % count];
        dst[i] = mulhi64(splitmix64(seed, i), q);
    }
}

// NTT of coeff * X^degree: out[i] = coeff * psi^((2*brv(i)+1)*degree mod 2N)  (table.rs:565-609)
__global__ __launch_bounds__(kPwThreads) void monomial_kernel(u64 *__restrict__ out,
                                                              const NttPrime *__restrict__ primes, u32 L,
                                                              u32 log_n, u64 degree, MonomialScalars sc) {
    const u64 n = 1ull << log_n;
    const u64 total = n * L;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u32 limb = (u32)(t >> log_n);
        const u32 i = (u32)(t & (n - 1));
        const NttPrime *P = primes + limb;
        const u32 r = log_n == 0 ? 0u : (__brev(i) >> (32 - log_n));
        const u64 idx = ((2ull * r + 1) * degree) & (2 * n - 1);
        const u32 k = (u32)(idx & (n - 1));
        const u32 kb = log_n == 0 ? 0u : (__brev(k) >> (32 - log_n));
        u64 w = P->fwd[kb].x;         // psi^k
        if (idx >= n) w = P->q - w;   // psi^(k+N) = -psi^k
        out[t] = mul_shoup(w, sc.value[limb], sc.quotient[limb], P->q);
    }
}

u32 grid_for(u64 work_items) {
    u64 g = (work_items + kPwThreads - 1) / kPwThreads;
    constexpr unsigned long long kPwWgPerCu = 1ull << 22;
    const u64 cap = 256ull * kPwWgPerCu < 0x7fffffffull ? 256ull * kPwWgPerCu : 0x7fffffffull;  // grid-stride beyond that
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (u32)g;
}

}  // namespace

int pointwise_dev(u64 *out, const u64 *a, const u64 *b, const u64 *c, const NttPrime *primes, u32 L, u32 log_n,
                  u64 len, u64 len_b, hipStream_t s, u64 group_words, bool pm) {
    if (len == 0) return PFHE_OK;
    const bool pair = log_n >= 1 && (len % 2 == 0) && (len_b % 2 == 0);
    const u64 items = ((pair ? len / 2 : len) + kPwUnroll - 1) / kPwUnroll;  // vectors per thread and iteration
    const dim3 g(grid_for(items ? items : 1)), t(kPwThreads);
#define PFHE_PW_LAUNCH(HAS_C, PAIR, PM)                                                                               \
    hipLaunchKernelGGL((pointwise_kernel<HAS_C, PAIR, PM>), g, t, 0, s, out, a, b, c, primes, L, log_n, len, len_b, \
                       group_words)
    const int variant = (c != nullptr ? 4 : 0) | (pair ? 2 : 0) | (pm ? 1 : 0);
    switch (variant) {
        case 0: PFHE_PW_LAUNCH(false, false, false); break;
        case 1: PFHE_PW_LAUNCH(false, false, true); break;
        case 2: PFHE_PW_LAUNCH(false, true, false); break;
        case 3: PFHE_PW_LAUNCH(false, true, true); break;
        case 4: PFHE_PW_LAUNCH(true, false, false); break;
        case 5: PFHE_PW_LAUNCH(true, false, true); break;
        case 6: PFHE_PW_LAUNCH(true, true, false); break;
        default: PFHE_PW_LAUNCH(true, true, true); break;
    }
#undef PFHE_PW_LAUNCH
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int butterfly_dev(bool factor, u64 *a, const u64 *s, const u64 *w, u64 *b, const NttPrime *primes, u32 L, u32 log_n,
                  u64 len, u64 len_w, hipStream_t st, bool pm) {
    if (len == 0) return PFHE_OK;
    const bool shared = len_w != len;  // len_w counts multiplicands (a factor is two words)
    const bool pair = log_n >= 1;
    const dim3 g(grid_for(pair ? len / 2 : len)), t(kPwThreads);
#define PFHE_BF_LAUNCH(F, PR, PMV) \
    hipLaunchKernelGGL((butterfly_kernel<F, PR, PMV>), g, t, 0, st, a, s, w, b, primes, L, log_n, len, shared)
    if (factor) {
        if (pair) PFHE_BF_LAUNCH(true, true, false); else PFHE_BF_LAUNCH(true, false, false);
    } else if (pm) {
        if (pair) PFHE_BF_LAUNCH(false, true, true); else PFHE_BF_LAUNCH(false, false, true);
    } else {
        if (pair) PFHE_BF_LAUNCH(false, true, false); else PFHE_BF_LAUNCH(false, false, false);
    }
#undef PFHE_BF_LAUNCH
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int fill_uniform_dev(u64 *dst, u64 len, const u64 *moduli_dev, u64 count, u64 poly_len, u64 seed,
                     hipStream_t s) {
    if (len == 0) return PFHE_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(len)), dim3(kPwThreads), 0, s, dst, len, moduli_dev,
                       count, poly_len, seed);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int monomial_dev(u64 *out, const NttPrime *primes, u32 L, u32 log_n, u64 degree, const MonomialScalars &sc,
                 hipStream_t s) {
    const u64 total = ((u64)L) << log_n;
    hipLaunchKernelGGL(monomial_kernel, dim3(grid_for(total)), dim3(kPwThreads), 0, s, out, primes, L, log_n,
                       degree, sc);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe
