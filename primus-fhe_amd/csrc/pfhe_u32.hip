// pfhe_u32.hip — the u32 / low-q tables: U32NttTable (primus_ntt/src/ntt/prime32/table.rs) and
// U32DcrtTable (primus_ntt/src/dcrt/prime32.rs) behind the C ABI (include/pfhe.h, "u32 tables").
//
// The transforms run the same strided / block kernels as the 64-bit path, instantiated with
// B32Arith (pfhe_ntt_device.hpp): a 64-bit word carries two adjacent u32 coefficients, so a
// polynomial of N coefficients is transformed as N/2 words plus one intra-word stage.  This file
// holds what is specific to the u32 tables: table construction in the packed layout, the
// streaming kernels on u32 data (pointwise products, monomial transforms, synthetic fill) and the
// extern "C" entry points.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "pfhe_capi_internal.hpp"
#include "pfhe_common.hpp"
#include "pfhe_handles.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"
#include "pfhe_pointwise.hpp"
#include "pfhe_staging.hpp"

namespace pfhe {

namespace {

constexpr int kThreads = 256;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

u32 grid_for(u64 items) {
    u64 g = (items + kThreads - 1) / kThreads;
    const u64 cap = 0x7fffffffull;  // one workgroup per 256 items streams fastest (see pfhe_elementwise.hip)
    if (g > cap) g = cap;
    return (u32)(g ? g : 1);
}

// x mod q for x < 2^62, q < 2^30; bar = floor(2^64 / q).  The estimate floor(x*bar / 2^64) is the
// true quotient or one less.
__device__ __forceinline__ u32 red64(u64 x, u32 q, u64 bar) {
    const u64 r = x - mulhi64(x, bar) * q;
    return (u32)(r >= q ? r - q : r);
}

// MODE 0: acc = acc*b; MODE 1: acc = a*b + acc — BarrettModulus<u32>::reduce_mul / reduce_mul_add
// on every limb (primus_modulus/src/barrett/ops.rs), canonical in and out.
template <int MODE>
__global__ __launch_bounds__(kThreads) void pointwise32_kernel(u32 *acc, const u32 *a, const u32 *__restrict__ b,
                                                               const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                               u64 len, u64 len_b) {
    const u64 nvec = len >> 2;
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (u64)gridDim.x * blockDim.x) {
        const u64 i = v << 2;
        const NttPrime *P = primes + (u32)((i >> log_n) % L);
        const u32 q = (u32)P->q;
        const u64 bar = P->bar_lo;
        const u64 ib = len_b != len ? i % len_b : i;
        const u32x4 x = *reinterpret_cast<const u32x4 *>((MODE == 0 ? acc : a) + i);
        const u32x4 y = *reinterpret_cast<const u32x4 *>(b + ib);
        u32x4 r;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = red64((u64)x[e] * y[e], q, bar);
        } else {
            const u32x4 z = *reinterpret_cast<const u32x4 *>(acc + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = red64((u64)x[e] * y[e] + z[e], q, bar);
        }
        *reinterpret_cast<u32x4 *>(acc + i) = r;
    }
}

// scalar form for polynomials shorter than one vector (N < 4)
template <int MODE>
__global__ void pointwise32_small_kernel(u32 *acc, const u32 *a, const u32 *__restrict__ b,
                                         const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 len, u64 len_b) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (u64)gridDim.x * blockDim.x) {
        const NttPrime *P = primes + (u32)((i >> log_n) % L);
        const u64 x = (u64)(MODE == 0 ? acc : a)[i] * b[len_b != len ? i % len_b : i] + (MODE == 1 ? acc[i] : 0u);
        acc[i] = red64(x, (u32)P->q, P->bar_lo);
    }
}

// NTT of coeff * X^degree (table.rs:376-470): out[i] = coeff * psi^((2*brv(i)+1)*degree mod 2N);
// psi^k for k < N is the low half of the packed forward table at brv(k), psi^(k+N) = -psi^k.
__global__ __launch_bounds__(kThreads) void monomial32_kernel(u32 *__restrict__ out,
                                                              const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                              u64 degree, MonomialScalars sc) {
    const u64 n = 1ull << log_n;
    const u64 total = n * L;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u32 limb = (u32)(t >> log_n);
        const u32 i = (u32)(t & (n - 1));
        const NttPrime *P = primes + limb;
        const u32 q = (u32)P->q;
        const u32 r = log_n == 0 ? 0u : (__brev(i) >> (32 - log_n));
        const u64 idx = ((2ull * r + 1) * degree) & (2 * n - 1);
        const u32 k = (u32)(idx & (n - 1));
        const u32 kb = log_n == 0 ? 0u : (__brev(k) >> (32 - log_n));
        u32 w = (u32)P->fwd_w[kb];
        if (idx >= n) w = q - w;
        out[t] = red64((u64)w * (u32)sc.value[limb], q, P->bar_lo);
    }
}

__device__ __forceinline__ u64 splitmix64(u64 seed, u64 i) {
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kThreads) void fill_uniform32_kernel(u32 *__restrict__ dst, u64 len,
                                                                  const NttPrime *__restrict__ primes, u32 L,
                                                                  u32 log_n, u64 seed) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (u64)gridDim.x * blockDim.x) {
        const u64 q = primes[(u32)((i >> log_n) % L)].q;
        dst[i] = (u32)mulhi64(splitmix64(seed, i), q);
    }
}

int check_len32(const TableSet &t, size_t len, u64 &units) {
    const size_t unit = t.n * t.L;
    if (len % unit != 0) {
        set_last_error("slice length is not a multiple of the polynomial length");
        return PFHE_ERR_BAD_LENGTH;
    }
    units = len / unit;
    return PFHE_OK;
}

}  // namespace

// U32NttTable::new for every modulus (table.rs:184-333), uploaded in the packed layout B32Arith
// reads: one 64-bit entry {w, floor(w*2^32/q)} per twiddle.
int make_table_set32(u32 log_n, const u32 *moduli, size_t count, int device, std::unique_ptr<TableSet> &out) {
    if (count == 0 || !moduli) {
        set_last_error("empty modulus list");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    std::vector<HostTable> host(count);
    for (size_t i = 0; i < count; ++i) {
        // root search first (table.rs:189), then the q < 2^30 requirement (:195-200)
        PFHE_TRY(build_host_table(log_n, moduli[i], host[i]));
        if (moduli[i] >= (1u << 30)) {
            set_last_error("modulus is too large for a u32 NTT table (max 30 bits)");
            return PFHE_ERR_MODULUS_TOO_LARGE;
        }
    }
    PFHE_TRY(capi_check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;

    auto ts = std::make_unique<TableSet>();
    ts->device = device;
    ts->log_n = log_n;
    ts->n = (size_t)1 << log_n;
    ts->L = (u32)count;
    ts->tune = NttTuning::from_env();  // u32 tables read the tuning switches at creation too
    ts->primes.resize(count);
    const size_t n = ts->n;
    for (size_t i = 0; i < count; ++i) {
        const u64 q = host[i].q;
        NttPrime &P = ts->primes[i];
        std::memset(&P, 0, sizeof P);
        P.q = q;
        P.two_q = q << 1;
        P.inv_n = host[i].inv_n;
        P.inv_n_p = (host[i].inv_n << 32) / q;
        P.inv_n_w = host[i].inv_n_w;
        P.inv_n_w_p = (host[i].inv_n_w << 32) / q;
        P.bar_lo = (u64)(((unsigned __int128)1 << 64) / q);
        const auto upload = [&](const std::vector<u64> &v, const u64 **dst) -> int {
            void *d = nullptr;
            PFHE_HIP(counted_malloc(&d, v.size() * sizeof(u64)));
            ts->allocations.push_back(d);
            PFHE_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(u64), hipMemcpyHostToDevice));
            *dst = static_cast<const u64 *>(d);
            return PFHE_OK;
        };
        const auto pack_of = [&](const std::vector<ulonglong2> &src, bool negate) {
            std::vector<u64> v(n);
            for (size_t k = 0; k < n; ++k)
                v[k] = (negate ? (u64)(u32)(0u - (u32)src[k].x) : src[k].x) | (((src[k].x << 32) / q) << 32);
            return v;
        };
        // forward, inverse, forward with the twiddle negated (B32Arith::mul1_neg)
        const std::vector<u64> pf = pack_of(host[i].fwd, false), pi = pack_of(host[i].inv, false), pn = pack_of(host[i].fwd, true);
        const u64 *dinv = nullptr;
        PFHE_TRY(upload(pf, &P.fwd_w));
        PFHE_TRY(upload(pi, &dinv));
        P.inv_w = dinv + n / 2;  // the word kernels index the inverse table in units of words: biased by N/2 entries
        PFHE_TRY(upload(pn, &P.fwd_wn));
        // Lane-ordered copies for the register pass in which a thread owns 16 consecutive WORDS (stages at word
        // distances 8, 4, 2, 1: 15 twiddles per group of 16 words, as NttPrime::fwd_last) plus the intra-word stage (one
        // twiddle per word: 16 more): entry (slot * G + g) belongs to group g, so the 64 lanes of a wave — 64
        // consecutive groups — load 64 consecutive entries instead of 64 separate lines (the gathers kept the u32 block
        // pass at 3.6 TB/s whatever its instruction count).  Word units: nw = N/2 words per polynomial.
        const size_t nw = n / 2;
        if (nw >= 16) {
            const size_t G = nw / 16;
            std::vector<u64> fl(31 * G), il(31 * G);
            for (int j = 3; j >= 0; --j) {
                const size_t per = (size_t)8 >> j;
                for (size_t u = 0; u < per; ++u)
                    for (size_t g2 = 0; g2 < G; ++g2) {
                        const size_t off = (per - 1 + u) * G + g2;
                        fl[off] = pn[(nw >> (j + 1)) + g2 * per + u];
                        il[off] = pi[nw + 1 + nw - (nw >> j) + g2 * per + u];
                    }
            }
            for (size_t k = 0; k < 16; ++k)
                for (size_t g2 = 0; g2 < G; ++g2) {
                    fl[(15 + k) * G + g2] = pn[nw + 16 * g2 + k];  // fwd_intra: roots[N/2 + word]
                    il[(15 + k) * G + g2] = pi[1 + 16 * g2 + k];   // inv_intra: inv_roots[1 + word]
                }
            PFHE_TRY(upload(fl, &P.fwd_last_w));
            PFHE_TRY(upload(il, &P.inv_last_w));
        }
        ts->roots.push_back(host[i].root);
        ts->inv_roots.push_back(host[i].inv_root);
    }
    void *pd = nullptr;
    PFHE_HIP(counted_malloc(&pd, count * sizeof(NttPrime)));
    ts->allocations.push_back(pd);
    PFHE_HIP(hipMemcpy(pd, ts->primes.data(), count * sizeof(NttPrime), hipMemcpyHostToDevice));
    ts->primes_dev = static_cast<const NttPrime *>(pd);
    out = std::move(ts);
    return PFHE_OK;
}

namespace {

int transform32_dev(const TableSet &t, u32 *data, size_t len, bool inverse, bool lazy, hipStream_t s) {
    if (!data && len) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(data);
    u64 units = 0;
    PFHE_TRY(check_len32(t, len, units));
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return ntt32_transform_dev(t.primes_dev, t.L, t.log_n, data, units * t.L, inverse, lazy, s, t.tune);
}

// host-pointer form: pooled staging context; a slice in memory the caller pinned is pipelined in pieces of whole units over
// its two streams, pageable slices are copied as one piece (see transform_host in pfhe_capi.hip)
int transform32_host(const TableSet &t, u32 *host, size_t len, bool inverse, bool lazy) {
    if (!host && len) return PFHE_ERR_BAD_ARGUMENT;
    u64 units = 0;
    PFHE_TRY(check_len32(t, len, units));
    if (len == 0) return PFHE_OK;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(t.device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *dv = nullptr;
    PFHE_TRY(st.alloc(len * sizeof(u32), &dv));
    u32 *d = static_cast<u32 *>(dv);
    const size_t unit = t.n * t.L;
    const bool pinned = st.pin(host, len * sizeof(u32));
    const size_t per = pinned ? std::max<size_t>(1, stage_chunk_bytes() / (unit * sizeof(u32))) : (size_t)units;
    const bool pipelined = per < units;
    const hipStream_t s_in = st.stream(), s_run = pipelined ? st.stream2() : st.stream();
    for (u64 u0 = 0; u0 < units; u0 += per) {
        const size_t words = (size_t)std::min<u64>(per, units - u0) * unit, off = (size_t)u0 * unit;
        PFHE_TRY(st.copy_in(d + off, host + off, words * sizeof(u32), s_in));
        if (pipelined) PFHE_TRY(st.order(s_in, s_run));
        PFHE_TRY(transform32_dev(t, d + off, words, inverse, lazy, s_run));
        PFHE_TRY(st.download(host + off, d + off, words * sizeof(u32), s_run));
    }
    return st.finish();
}

int pointwise32(const TableSet &t, int mode, u32 *acc, const u32 *a, size_t len_a, const u32 *b, size_t len_b,
                hipStream_t s) {
    if ((!acc || !b || (mode == 1 && !a)) && len_a) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(acc);
    PFHE_REQUIRE_ALIGNED(a);
    PFHE_REQUIRE_ALIGNED(b);
    u64 units = 0;
    PFHE_TRY(check_len32(t, len_a, units));
    if (len_b != len_a && len_b != t.n * t.L) {
        set_last_error("multiplicand must have the same length or exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len_a == 0) return PFHE_OK;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    const u64 len = len_a;
    if (t.log_n >= 2) {
        const dim3 grid(grid_for(len / 4)), block(kThreads);
        if (mode == 0) hipLaunchKernelGGL(pointwise32_kernel<0>, grid, block, 0, s, acc, a, b, t.primes_dev, t.L, t.log_n, len, (u64)len_b);
        else hipLaunchKernelGGL(pointwise32_kernel<1>, grid, block, 0, s, acc, a, b, t.primes_dev, t.L, t.log_n, len, (u64)len_b);
    } else {
        const dim3 grid(grid_for(len)), block(kThreads);
        if (mode == 0) hipLaunchKernelGGL(pointwise32_small_kernel<0>, grid, block, 0, s, acc, a, b, t.primes_dev, t.L, t.log_n, len, (u64)len_b);
        else hipLaunchKernelGGL(pointwise32_small_kernel<1>, grid, block, 0, s, acc, a, b, t.primes_dev, t.L, t.log_n, len, (u64)len_b);
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// minus_one: limb i uses q_i - 1 (DcrtTable::transform_coeff_minus_one_monomial, dcrt/mod.rs:124-134)
int monomial32(const TableSet &t, u32 coeff, size_t degree, u32 *values, size_t len, bool host, hipStream_t s,
               bool minus_one = false) {
    if (!values) return PFHE_ERR_BAD_ARGUMENT;
    if (len != t.n * t.L) {
        set_last_error("monomial output must be exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    // scalars by value, at most kMaxMonomialLimbs per launch; wider bases take one launch per group of limbs
    std::vector<MonomialScalars> groups((t.L + kMaxMonomialLimbs - 1) / kMaxMonomialLimbs);
    for (u32 i = 0; i < t.L; ++i) {
        const u32 q = (u32)t.primes[i].q;
        const u32 ci = minus_one ? q - 1 : coeff;
        if (ci >= q) {
            set_last_error("monomial coefficient must be reduced modulo every modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        groups[i / kMaxMonomialLimbs].value[i % kMaxMonomialLimbs] = ci;
    }
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    const u64 deg = (u64)degree & (2 * (u64)t.n - 1);
    void *out_dev = values;
    std::unique_ptr<HostStage> st;
    if (host) {  // pooled staging context: no allocation in steady state, the caller's stream is not involved
        st = std::make_unique<HostStage>(t.device);
        if (!st->ok()) return PFHE_ERR_HIP;
        PFHE_TRY(st->alloc(len * sizeof(u32), &out_dev));
        s = st->stream();
    }
    hipError_t e = hipSuccess;
    for (size_t gi = 0; gi < groups.size() && e == hipSuccess; ++gi) {
        const u32 l0 = (u32)gi * kMaxMonomialLimbs, lg = std::min<u32>(kMaxMonomialLimbs, t.L - l0);
        hipLaunchKernelGGL(monomial32_kernel, dim3(grid_for((size_t)lg * t.n)), dim3(kThreads), 0, s,
                           static_cast<u32 *>(out_dev) + (size_t)l0 * t.n, t.primes_dev + l0, lg, t.log_n, deg, groups[gi]);
        e = hipGetLastError();
    }
    if (host && e == hipSuccess) {  // the device form is these launches (capturable); only the host form copies back and waits
        PFHE_TRY(st->download(values, out_dev, len * sizeof(u32)));
        return st->finish();
    }
    if (e != hipSuccess) return hip_fail(e, "monomial transform", __FILE__, __LINE__);
    return PFHE_OK;
}

}  // namespace
}  // namespace pfhe

using namespace pfhe;

struct pfhe_ntt32 {
    std::unique_ptr<TableSet> t;
};
struct pfhe_dcrt32 {
    std::unique_ptr<TableSet> t;
};
namespace pfhe {
const TableSet *capi_table32_of(const pfhe_dcrt32 *t) { return t->t.get(); }
}  // namespace pfhe

extern "C" {

/* ---------------------------- U32NttTable ---------------------------- */

int pfhe_ntt32_create(uint32_t log_n, uint32_t modulus, int device, pfhe_ntt32 **out) {
    PFHE_GUARD_BEGIN
    if (!out) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<TableSet> t;
    u32 q = modulus;
    PFHE_TRY(make_table_set32(log_n, &q, 1, device, t));
    *out = new pfhe_ntt32{std::move(t)};
    return PFHE_OK;
    PFHE_GUARD_END
}
void pfhe_ntt32_destroy(pfhe_ntt32 *table) { delete table; }
size_t pfhe_ntt32_poly_length(const pfhe_ntt32 *t) { return t ? t->t->n : 0; }
uint32_t pfhe_ntt32_log_n(const pfhe_ntt32 *t) { return t ? t->t->log_n : 0; }
uint32_t pfhe_ntt32_modulus(const pfhe_ntt32 *t) { return t ? (uint32_t)t->t->primes[0].q : 0; }
uint32_t pfhe_ntt32_root(const pfhe_ntt32 *t) { return t ? (uint32_t)t->t->roots[0] : 0; }
uint32_t pfhe_ntt32_inv_root(const pfhe_ntt32 *t) { return t ? (uint32_t)t->t->inv_roots[0] : 0; }
uint32_t pfhe_ntt32_inv_n(const pfhe_ntt32 *t) { return t ? (uint32_t)t->t->primes[0].inv_n : 0; }
int pfhe_ntt32_device(const pfhe_ntt32 *t) { return t ? t->t->device : -1; }

#define PFHE_SLICE32(NAME, PREFIX, INV, LAZY)                                 \
    int NAME(const PREFIX *table, uint32_t *data, size_t len) {               \
        PFHE_GUARD_BEGIN                                                      \
        if (!table) return PFHE_ERR_BAD_ARGUMENT;                             \
        return transform32_host(*table->t, data, len, INV, LAZY);             \
        PFHE_GUARD_END                                                        \
    }
PFHE_SLICE32(pfhe_ntt32_transform_slice, pfhe_ntt32, false, false)
PFHE_SLICE32(pfhe_ntt32_inverse_transform_slice, pfhe_ntt32, true, false)
PFHE_SLICE32(pfhe_ntt32_lazy_transform_slice, pfhe_ntt32, false, true)
PFHE_SLICE32(pfhe_ntt32_lazy_inverse_transform_slice, pfhe_ntt32, true, true)
PFHE_SLICE32(pfhe_dcrt32_transform_slice, pfhe_dcrt32, false, false)
PFHE_SLICE32(pfhe_dcrt32_inverse_transform_slice, pfhe_dcrt32, true, false)
PFHE_SLICE32(pfhe_dcrt32_lazy_transform_slice, pfhe_dcrt32, false, true)
PFHE_SLICE32(pfhe_dcrt32_lazy_inverse_transform_slice, pfhe_dcrt32, true, true)
#undef PFHE_SLICE32

int pfhe_ntt32_transform_monomial(const pfhe_ntt32 *table, uint32_t coeff, size_t degree, uint32_t *values,
                                  size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial32(*table->t, coeff, degree, values, len, true, nullptr);
    PFHE_GUARD_END
}
int pfhe_ntt32_transform_coeff_one_monomial(const pfhe_ntt32 *table, size_t degree, uint32_t *values, size_t len) {
    return pfhe_ntt32_transform_monomial(table, 1, degree, values, len);
}
int pfhe_ntt32_transform_coeff_minus_one_monomial(const pfhe_ntt32 *table, size_t degree, uint32_t *values,
                                                  size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial32(*table->t, 0, degree, values, len, true, nullptr, true);
    PFHE_GUARD_END
}

int pfhe_ntt32_transform_dev(const pfhe_ntt32 *table, uint32_t *poly_dev, size_t len, int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform32_dev(*table->t, poly_dev, len, false, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_ntt32_inverse_transform_dev(const pfhe_ntt32 *table, uint32_t *values_dev, size_t len, int lazy,
                                     void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform32_dev(*table->t, values_dev, len, true, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_ntt32_transform_monomial_dev(const pfhe_ntt32 *table, uint32_t coeff, size_t degree, uint32_t *values_dev,
                                      size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial32(*table->t, coeff, degree, values_dev, len, false, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_ntt32_mul_assign_dev(const pfhe_ntt32 *table, uint32_t *a_dev, size_t len_a, const uint32_t *b_dev,
                              size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise32(*table->t, 0, a_dev, nullptr, len_a, b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_ntt32_add_mul_assign_dev(const pfhe_ntt32 *table, uint32_t *acc_dev, const uint32_t *a_dev, size_t len_a,
                                  const uint32_t *b_dev, size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise32(*table->t, 1, acc_dev, a_dev, len_a, b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}

/* ---------------------------- U32DcrtTable ---------------------------- */

int pfhe_dcrt32_create(uint32_t log_n, const uint32_t *moduli, size_t moduli_count, int device, pfhe_dcrt32 **out) {
    PFHE_GUARD_BEGIN
    if (!out) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<TableSet> t;
    PFHE_TRY(make_table_set32(log_n, moduli, moduli_count, device, t));
    *out = new pfhe_dcrt32{std::move(t)};
    return PFHE_OK;
    PFHE_GUARD_END
}
void pfhe_dcrt32_destroy(pfhe_dcrt32 *table) { delete table; }
size_t pfhe_dcrt32_poly_length(const pfhe_dcrt32 *t) { return t ? t->t->n : 0; }
size_t pfhe_dcrt32_moduli_count(const pfhe_dcrt32 *t) { return t ? t->t->L : 0; }
size_t pfhe_dcrt32_crt_poly_length(const pfhe_dcrt32 *t) { return t ? t->t->n * t->t->L : 0; }
int pfhe_dcrt32_device(const pfhe_dcrt32 *t) { return t ? t->t->device : -1; }
uint32_t pfhe_dcrt32_modulus(const pfhe_dcrt32 *t, size_t i) { return (t && i < t->t->L) ? (uint32_t)t->t->primes[i].q : 0; }
uint32_t pfhe_dcrt32_root(const pfhe_dcrt32 *t, size_t i) { return (t && i < t->t->L) ? (uint32_t)t->t->roots[i] : 0; }

int pfhe_dcrt32_transform_monomial(const pfhe_dcrt32 *table, uint32_t coeff, size_t degree, uint32_t *values,
                                   size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial32(*table->t, coeff, degree, values, len, true, nullptr);
    PFHE_GUARD_END
}
int pfhe_dcrt32_transform_coeff_one_monomial(const pfhe_dcrt32 *table, size_t degree, uint32_t *values, size_t len) {
    return pfhe_dcrt32_transform_monomial(table, 1, degree, values, len);
}
int pfhe_dcrt32_transform_coeff_minus_one_monomial(const pfhe_dcrt32 *table, size_t degree, uint32_t *values,
                                                   size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial32(*table->t, 0, degree, values, len, true, nullptr, true);
    PFHE_GUARD_END
}

int pfhe_dcrt32_transform_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len, int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform32_dev(*table->t, poly_dev, len, false, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_dcrt32_inverse_transform_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len, int lazy,
                                      void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform32_dev(*table->t, poly_dev, len, true, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_dcrt32_mul_assign_dev(const pfhe_dcrt32 *table, uint32_t *a_dev, size_t len_a, const uint32_t *b_dev,
                               size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise32(*table->t, 0, a_dev, nullptr, len_a, b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_dcrt32_add_mul_assign_dev(const pfhe_dcrt32 *table, uint32_t *acc_dev, const uint32_t *a_dev, size_t len_a,
                                   const uint32_t *b_dev, size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise32(*table->t, 1, acc_dev, a_dev, len_a, b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt32_fill_uniform_dev(const pfhe_dcrt32 *table, uint32_t *dst_dev, size_t len, uint64_t seed,
                                 void *stream) {
    PFHE_GUARD_BEGIN
    if (!table || (!dst_dev && len)) return PFHE_ERR_BAD_ARGUMENT;
    if (len == 0) return PFHE_OK;
    const TableSet &t = *table->t;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    hipLaunchKernelGGL(fill_uniform32_kernel, dim3(grid_for(len)), dim3(kThreads), 0, (hipStream_t)stream, dst_dev,
                       (u64)len, t.primes_dev, t.L, t.log_n, seed);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

/* profiling hooks: the passes of one transform, as for the 64-bit tables */
int pfhe_dcrt32_transform_num_passes(const pfhe_dcrt32 *table) {
    if (!table) return 0;
    return table->t->log_n <= 4 ? 1 : ntt_num_passes(table->t->log_n - 1, kArithB32, table->t->tune);
}
const char *pfhe_dcrt32_transform_pass_name(const pfhe_dcrt32 *table, int inverse, int index) {
    static thread_local char buf[112];
    buf[0] = 0;
    if (!table) return buf;
    if (table->t->log_n <= 4) {
        std::snprintf(buf, sizeof buf, "ntt32_tiny_kernel");
        return buf;
    }
    char inner[96];
    ntt_pass_name(table->t->log_n - 1, inverse != 0, index, inner, sizeof inner, kArithB32, table->t->tune);
    std::snprintf(buf, sizeof buf, "u32:%s", inner);
    return buf;
}
int pfhe_dcrt32_transform_form(const pfhe_dcrt32 *table, size_t len, int inverse, char *name, size_t cap, int *launches) {
    if (!table || !name || cap == 0 || !launches) return PFHE_ERR_BAD_ARGUMENT;
    const TableSet &t = *table->t;
    if (len % (t.n * t.L) != 0) return PFHE_ERR_BAD_LENGTH;
    if (t.log_n <= 4) {
        std::snprintf(name, cap, "ntt32_tiny_kernel");
        *launches = 1;
        return PFHE_OK;
    }
    char inner[96];
    *launches = ntt_transform_form(t.L, t.log_n - 1, kArithB32, len / t.n, inverse != 0, t.tune, inner, sizeof inner);
    std::snprintf(name, cap, "u32:%s", inner);
    return PFHE_OK;
}
int pfhe_dcrt32_transform_pass_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len, int inverse, int index,
                                   int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table || (!poly_dev && len)) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(poly_dev);
    const TableSet &t = *table->t;
    if (len % (t.n * t.L) != 0) return PFHE_ERR_BAD_LENGTH;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    if (t.log_n <= 4) {
        if (index != 0) return PFHE_ERR_BAD_ARGUMENT;
        return ntt32_transform_dev(t.primes_dev, t.L, t.log_n, poly_dev, len / t.n, inverse != 0, lazy != 0,
                                   (hipStream_t)stream);
    }
    return ntt_pass_dev(t.primes_dev, t.L, t.log_n - 1, kArithB32, reinterpret_cast<u64 *>(poly_dev), len / t.n,
                        inverse != 0, index, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}

}  // extern "C"
