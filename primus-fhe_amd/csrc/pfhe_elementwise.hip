// pfhe_elementwise.hip — the CrtPolynomial / DcrtPolynomial / CrtGlwe element-wise family that sits either
// side of the transforms and the external product.  Canonical residues in, canonical residues out; every
// kernel is a stream over HBM (16-byte accesses, per-limb constants selected from the element index).
//
//   add / sub / neg           primus_poly/src/crt/{add,sub,neg}.rs, dcrt/{add,sub,neg}.rs;
//                             CrtGlwe::{add,sub}_element_wise* (primus_lattice/src/macros/mod.rs:367-531)
//                             per limb: compact::reduce_add / reduce_sub / reduce_neg
//                             (primus_modulus/src/common/compact/primitive.rs:10-46, common/uint/primitive.rs:19-33)
//   mul_scalar, add_mul_scalar  crt/mul.rs:16-77,138-158, dcrt/mul.rs:78-119,254-275 -> BarrettModulus::reduce_mul /
//                             reduce_mul_add (primus_modulus/src/barrett/ops.rs:276-315)
//   mul_factor, add_mul_factor  crt/mul.rs:37-99,161-180 -> ShoupFactor::factor_mul_modulo + reduce_add
//                             (primus_factor/src/shoup_factor/mod.rs:124-143, common/slice.rs:7-70)
//   mul_monomial              crt/mul.rs:102-127, CrtGlwe::mul_monic_monomial_assign (glwe/crt.rs:76-113):
//                             rotate_right(r) + negate the wrapped part
//   inv                       dcrt/inv.rs:19-68 -> Montgomery batch inversion (barrett/slice.rs:505-558); here each
//                             thread inverts the product of 16 strided elements of one limb polynomial
#include <algorithm>
#include <cstdlib>

#include "pfhe_capi_internal.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_staging.hpp"
#include "../../include/pfhe.h"

namespace pfhe {

namespace {

constexpr int kEwThreads = 256;
constexpr int kMaxEwLimbs = 32;  // per-limb scalars travel as kernel arguments (512 bytes)
// Launch shape, measured on 3 GiB operands (tools/perf_elementwise.py): one 16-byte vector per thread and as many
// workgroups as there are vectors (the dispatcher then walks memory in address order) with non-temporal loads and
// stores streams at 6.3 TB/s; the usual "8 workgroups per CU + grid-stride loop, 4 vectors in flight per thread"
// reaches 4.8 TB/s (5.3 TB/s with non-temporal accesses).  The grid-stride loop remains for > 2^31 workgroups.
constexpr int kEwUnroll = 1;                // vectors per thread and iteration
constexpr unsigned kEwWgPerCu = 1u << 22;   // workgroups per CU before the grid-stride loop takes over

using ew_vec = __attribute__((__vector_size__(2 * sizeof(u64)))) u64;
__device__ __forceinline__ ew_vec ew_load(const u64 *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const ew_vec *>(p));
}
__device__ __forceinline__ void ew_store(u64 *p, ew_vec v) {
    __builtin_nontemporal_store(v, reinterpret_cast<ew_vec *>(p));
}

enum EwOp : int { kAdd, kSub, kNeg, kMulScalar, kAddMulScalar, kMulFactor, kAddMulFactor };

// per-limb scalars passed by value (kernel arguments live in SGPRs / constant memory)
struct EwScalars {
    u64 value[kMaxEwLimbs];
    u64 quotient[kMaxEwLimbs];
};

__device__ __forceinline__ u64 neg_mod(u64 x, u64 q) { return x ? q - x : 0; }

template <int OP>
__device__ __forceinline__ u64 ew_apply(u64 a, u64 b, const NttPrime &P, u64 sv, u64 sq) {
    if constexpr (OP == kAdd) return add_mod(a, b, P.q);
    else if constexpr (OP == kSub) return sub_mod(a, b, P.q);
    else if constexpr (OP == kNeg) return neg_mod(a, P.q);
    // a scalar is a factor whose quotient the host computed: reduce_mul / reduce_mul_add and the Shoup product
    // all return THE canonical residue, so the cheaper multiply is bit-identical
    else if constexpr (OP == kMulScalar || OP == kMulFactor) return mul_shoup(a, sv, sq, P.q);
    else if constexpr (OP == kAddMulScalar) return add_mod(a, mul_shoup(b, sv, sq, P.q), P.q);
    else return add_mod(a, mul_shoup(b, sv, sq, P.q), P.q);
}

template <int OP>
constexpr bool ew_has_b() {
    return OP == kAdd || OP == kSub || OP == kAddMulScalar || OP == kAddMulFactor;
}

// out[i] = op(a[i], b[i]); out may alias a and/or b.  UNROLL independent 16-byte vectors per thread and
// iteration, one grid-stride apart.
template <int OP, bool PAIR>
__global__ __launch_bounds__(kEwThreads) void elementwise_kernel(u64 *out, const u64 *a, const u64 *b,
                                                                 const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                                 u64 len, EwScalars sc) {
    constexpr u64 V = PAIR ? 2 : 1;
    constexpr int UNROLL = kEwUnroll;
    const u64 nvec = len / V;
    const u64 tile = (u64)gridDim.x * blockDim.x;
    for (u64 v0 = (u64)blockIdx.x * blockDim.x + threadIdx.x; v0 < nvec; v0 += tile * UNROLL) {
        u64 av[UNROLL][2], bv[UNROLL][2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 v = v0 + tile * u;
            if (v >= nvec) continue;
            const u64 i = v * V;
            if constexpr (PAIR) {
                const ew_vec x = ew_load(a + i);
                av[u][0] = x[0]; av[u][1] = x[1];
                if constexpr (ew_has_b<OP>()) {
                    const ew_vec y = ew_load(b + i);
                    bv[u][0] = y[0]; bv[u][1] = y[1];
                }
            } else {
                av[u][0] = a[i];
                if constexpr (ew_has_b<OP>()) bv[u][0] = b[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u64 v = v0 + tile * u;
            if (v >= nvec) continue;
            const u64 i = v * V;
            u32 p = (u32)(i >> log_n);  // uniform across a wave once a polynomial spans its 128 words
            if (log_n >= 7) p = __builtin_amdgcn_readfirstlane(p);
            const u32 limb = p % L;
            const NttPrime &P = primes[limb];
            u64 r[2];
#pragma unroll
            for (int e = 0; e < (int)V; ++e)
                r[e] = ew_apply<OP>(av[u][e], ew_has_b<OP>() ? bv[u][e] : 0, P, sc.value[limb], sc.quotient[limb]);
            if constexpr (PAIR) ew_store(out + i, ew_vec{r[0], r[1]});
            else out[i] = r[0];
        }
    }
}

// out[j] = +-in[(j - r) mod N] within every N-word polynomial (out != in): for r < N the first r outputs are
// the negated wrap-around, for r = N + r' every output but the first r' is negated (crt/mul.rs:102-127).
// A thread writes one 16-byte vector; its two source words are adjacent too (but not 16-byte aligned for odd r).
template <bool PAIR>
__global__ __launch_bounds__(kEwThreads) void monomial_rotate_kernel(u64 *__restrict__ out, const u64 *__restrict__ in,
                                                                     const NttPrime *__restrict__ primes, u32 L,
                                                                     u32 log_n, u64 len, u32 rot, bool high) {
    constexpr u64 V = PAIR ? 2 : 1;
    const u64 nvec = len / V;
    const u32 mask = (1u << log_n) - 1;
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (u64)gridDim.x * blockDim.x) {
        const u64 i = v * V;
        const u64 base = (i >> log_n) << log_n;
        const u32 j = (u32)(i - base);
        const u64 q = primes[(u32)((i >> log_n) % L)].q;
        u64 r[2];
#pragma unroll
        for (int e = 0; e < (int)V; ++e) {
            const u32 je = j + e;
            const u64 x = __builtin_nontemporal_load(in + base + ((je - rot) & mask));
            const bool negate = (je < rot) != high;
            r[e] = negate ? neg_mod(x, q) : x;
        }
        if constexpr (PAIR) ew_store(out + i, ew_vec{r[0], r[1]});
        else out[i] = r[0];
    }
}

// In-place X^r for rings that fit one workgroup's registers (2^9 <= N <= 2^14): a workgroup owns one N-word
// polynomial, every thread loads its WPT words (lane-consecutive 8-byte accesses), the workgroup waits until all
// of them have arrived, and only then are the rotated (and, past the wrap, negated) words written back.
template <int WPT>
__global__ __launch_bounds__(1024) void monomial_inplace_kernel(u64 *__restrict__ data, const NttPrime *__restrict__ primes,
                                                               u32 L, u32 log_n, u32 rot, bool high) {
    const u32 n = 1u << log_n, mask = n - 1, threads = n / WPT;
    const u64 poly = blockIdx.x;
    u64 *__restrict__ base = data + poly * n;
    const u64 q = primes[(u32)(poly % L)].q;
    const u32 lt = threadIdx.x;
    u64 v[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) v[j] = __builtin_nontemporal_load(base + lt + threads * j);
    // the loads must have RETURNED (not merely issued) before any wave of the workgroup overwrites their addresses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const u32 i = lt + threads * j, d = i + rot;
        const bool negate = (d >= n) != high;
        __builtin_nontemporal_store(negate ? neg_mod(v[j], q) : v[j], base + (d & mask));
    }
}

// Point-wise inverse.  Thread t of polynomial p owns the E elements p*N + e*(N/E) + t: prefix products,
// ONE Fermat inversion of the total, back-substitution — the reference's batch inversion (barrett/slice.rs:
// 505-558) with a batch of E per thread; every output is the unique inverse, so the grouping is not visible.
// A zero element (no inverse; the reference panics) raises *flag and leaves that thread's outputs unspecified.
template <int E>
__global__ __launch_bounds__(kEwThreads) void inv_kernel(u64 *out, const u64 *in, const NttPrime *__restrict__ primes,
                                                         u32 L, u32 log_n, u64 len, unsigned int *flag) {
    const u64 n = 1ull << log_n;
    const u64 per_poly = n / E;  // threads per polynomial
    const u64 threads = len / E;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < threads; g += (u64)gridDim.x * blockDim.x) {
        const u64 p = g / per_poly, t = g - p * per_poly;
        const NttPrime &P = primes[(u32)(p % L)];
        const u64 q = P.q, lo = P.bar_lo, hi = P.bar_hi;
        const u64 base = p * n + t;
        u64 x[E], pre[E];
        u64 total = 1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e] = in[base + (u64)e * per_poly];
            pre[e] = total;
            total = mul_mod_barrett(total, x[e], q, lo, hi);
        }
        if (total == 0) {
            atomicOr(flag, 1u);
            continue;
        }
        // total^(q-2) mod q, left-to-right square and multiply
        u64 inv = 1;
        const u64 ex = q - 2;
        for (int bit = 63 - __clzll((long long)ex); bit >= 0; --bit) {
            inv = mul_mod_barrett(inv, inv, q, lo, hi);
            if ((ex >> bit) & 1) inv = mul_mod_barrett(inv, total, q, lo, hi);
        }
#pragma unroll
        for (int e = E - 1; e >= 0; --e) {
            out[base + (u64)e * per_poly] = mul_mod_barrett(pre[e], inv, q, lo, hi);
            inv = mul_mod_barrett(inv, x[e], q, lo, hi);
        }
    }
}

// Plain copy in the launch shape of the element-wise family (one 16-byte vector per thread, non-temporal both ways): the
// streaming rate this library's own access pattern reaches, measured by bench.py as `device_copy` / `roofline.peak_measured`.
__global__ __launch_bounds__(kEwThreads) void stream_copy_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src, u64 nvec) {
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (u64)gridDim.x * blockDim.x)
        ew_store(dst + 2 * v, ew_load(src + 2 * v));
}

u32 ew_grid(u64 items) {
    u64 g = (items + kEwThreads - 1) / kEwThreads;
    const u64 cap = std::min<u64>(256ull * kEwWgPerCu, 0x7fffffffull);  // grid-stride beyond that
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (u32)g;
}

template <int OP>
int launch_elementwise(const TableSet &t, u64 *out, const u64 *a, const u64 *b, u64 len, const EwScalars &sc,
                       hipStream_t s) {
    if (len == 0) return PFHE_OK;
    const bool pair = t.log_n >= 1;
    const u64 items = ((pair ? len / 2 : len) + kEwUnroll - 1) / kEwUnroll;
    const dim3 g(ew_grid(items)), th(kEwThreads);
    if (pair) hipLaunchKernelGGL((elementwise_kernel<OP, true>), g, th, 0, s, out, a, b, t.primes_dev, t.L, t.log_n, len, sc);
    else hipLaunchKernelGGL((elementwise_kernel<OP, false>), g, th, 0, s, out, a, b, t.primes_dev, t.L, t.log_n, len, sc);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int check_common(const pfhe_dcrt *table, const void *p0, const void *p1, const void *p2, size_t len) {
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    const TableSet &t = *capi_table_of(table);
    if (len && (!p0 || !p1 || !p2)) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(p0);
    PFHE_REQUIRE_ALIGNED(p1);
    PFHE_REQUIRE_ALIGNED(p2);
    if (len % (t.n * t.L) != 0) {
        set_last_error("length must be a whole number of RNS polynomials (L * N words)");
        return PFHE_ERR_BAD_LENGTH;
    }
    return PFHE_OK;
}

template <int OP>
int binary_op(const pfhe_dcrt *table, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, void *stream) {
    if (int st = check_common(table, a, b, out, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    return launch_elementwise<OP>(*capi_table_of(table), (u64 *)out, (const u64 *)a, (const u64 *)b, len, EwScalars{}, (hipStream_t)stream);
}

// scalars: L plain residues (factor == false) or L (value, quotient) pairs
int load_scalars(const TableSet &t, const uint64_t *scalars, bool factor, EwScalars &sc) {
    if (!scalars) return PFHE_ERR_BAD_ARGUMENT;
    if (t.L > (u32)kMaxEwLimbs) {  // add / sub / neg / monomial / inv take any number of limbs; the scalar forms 32
        set_last_error("per-limb scalars are supported for at most 32 limbs");
        return PFHE_ERR_UNSUPPORTED;
    }
    for (u32 r = 0; r < t.L; ++r) {
        sc.value[r] = factor ? scalars[2 * r] : scalars[r];
        if (sc.value[r] >= t.primes[r].q) {
            set_last_error("scalar must be reduced modulo its limb's modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        sc.quotient[r] = factor ? scalars[2 * r + 1]
                                : (u64)((((unsigned __int128)sc.value[r]) << 64) / t.primes[r].q);
    }
    return PFHE_OK;
}

int monomial_to(const TableSet &t, const u64 *in, u64 r, u64 *out, u64 len, hipStream_t s) {
    if (len == 0) return PFHE_OK;
    const bool high = r >= t.n;
    const u32 rot = (u32)(high ? r - t.n : r);
    const bool pair = t.log_n >= 1;
    const dim3 g(ew_grid(pair ? len / 2 : len)), th(kEwThreads);
    if (pair) hipLaunchKernelGGL(monomial_rotate_kernel<true>, g, th, 0, s, out, in, t.primes_dev, t.L, t.log_n, len, rot, high);
    else hipLaunchKernelGGL(monomial_rotate_kernel<false>, g, th, 0, s, out, in, t.primes_dev, t.L, t.log_n, len, rot, high);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace

}  // namespace pfhe

using namespace pfhe;

extern "C" {

int pfhe_stream_copy_dev(int device, void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    PFHE_GUARD_BEGIN
    if (bytes == 0) return PFHE_OK;
    if (!dst_dev || !src_dev || bytes % 16 != 0) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(dst_dev);
    PFHE_REQUIRE_ALIGNED(src_dev);
    PFHE_TRY(capi_check_device(device));
    DeviceGuard guard(device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    const u64 nvec = bytes / 16;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(ew_grid(nvec)), dim3(kEwThreads), 0, (hipStream_t)stream, (u64 *)dst_dev,
                       (const u64 *)src_dev, nvec);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_dcrt_add_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev,
                         size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    return binary_op<kAdd>(table, a_dev, b_dev, out_dev, len, stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_sub_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev,
                         size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    return binary_op<kSub>(table, a_dev, b_dev, out_dev, len, stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_neg_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, uint64_t *out_dev, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, a_dev, a_dev, out_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    return launch_elementwise<kNeg>(*capi_table_of(table), (u64 *)out_dev, (const u64 *)a_dev, nullptr, len, EwScalars{},
                                    (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_scalar_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *scalars,
                                uint64_t *out_dev, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, a_dev, a_dev, out_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    EwScalars sc{};
    if (int st = load_scalars(*capi_table_of(table), scalars, false, sc)) return st;
    return launch_elementwise<kMulScalar>(*capi_table_of(table), (u64 *)out_dev, (const u64 *)a_dev, nullptr, len, sc,
                                          (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_add_mul_scalar_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev, const uint64_t *rhs_dev,
                                        const uint64_t *scalars, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, acc_dev, rhs_dev, acc_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    EwScalars sc{};
    if (int st = load_scalars(*capi_table_of(table), scalars, false, sc)) return st;
    return launch_elementwise<kAddMulScalar>(*capi_table_of(table), (u64 *)acc_dev, (const u64 *)acc_dev, (const u64 *)rhs_dev, len,
                                             sc, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_factor_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *factors,
                                uint64_t *out_dev, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, a_dev, a_dev, out_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    EwScalars sc{};
    if (int st = load_scalars(*capi_table_of(table), factors, true, sc)) return st;
    return launch_elementwise<kMulFactor>(*capi_table_of(table), (u64 *)out_dev, (const u64 *)a_dev, nullptr, len, sc,
                                          (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_add_mul_factor_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev, const uint64_t *rhs_dev,
                                        const uint64_t *factors, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, acc_dev, rhs_dev, acc_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    EwScalars sc{};
    if (int st = load_scalars(*capi_table_of(table), factors, true, sc)) return st;
    return launch_elementwise<kAddMulFactor>(*capi_table_of(table), (u64 *)acc_dev, (const u64 *)acc_dev, (const u64 *)rhs_dev, len,
                                             sc, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_monomial_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t r, uint64_t *out_dev,
                                  size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, a_dev, a_dev, out_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    const TableSet &t = *capi_table_of(table);
    if (r >= 2 * t.n) {
        set_last_error("monomial degree must be below 2N");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    // the rotation reads in[(j - r) mod N] while other threads write out[j]: ANY overlap of the two ranges is a race
    if (len) {
        const uintptr_t a0 = (uintptr_t)a_dev, o0 = (uintptr_t)out_dev, bytes = (uintptr_t)len * sizeof(u64);
        if (a0 < o0 + bytes && o0 < a0 + bytes) {
            set_last_error("mul_monomial_to needs non-overlapping buffers; use mul_monomial_assign for the in-place form");
            return PFHE_ERR_BAD_ARGUMENT;
        }
    }
    return monomial_to(t, (const u64 *)a_dev, r, (u64 *)out_dev, len, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_monomial_assign_dev(const pfhe_dcrt *table, uint64_t *data_dev, size_t r, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, data_dev, data_dev, data_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    const TableSet &t = *capi_table_of(table);
    if (r >= 2 * t.n) {
        set_last_error("monomial degree must be below 2N");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    if (len == 0) return PFHE_OK;
    if (t.log_n >= 9 && t.log_n <= 14 && len / t.n <= 0x7fffffffull) {
        // truly in place: one workgroup per polynomial, data held in registers across the barrier
        const bool high = r >= t.n;
        const u32 rot = (u32)(high ? r - t.n : r);
        const dim3 g((u32)(len / t.n));
        u64 *d = (u64 *)data_dev;
        hipStream_t st = (hipStream_t)stream;
        switch (t.log_n) {
            case 9: hipLaunchKernelGGL(monomial_inplace_kernel<2>, g, dim3(256), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
            case 10: hipLaunchKernelGGL(monomial_inplace_kernel<4>, g, dim3(256), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
            case 11: hipLaunchKernelGGL(monomial_inplace_kernel<8>, g, dim3(256), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
            case 12: hipLaunchKernelGGL(monomial_inplace_kernel<16>, g, dim3(256), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
            case 13: hipLaunchKernelGGL(monomial_inplace_kernel<16>, g, dim3(512), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
            default: hipLaunchKernelGGL(monomial_inplace_kernel<16>, g, dim3(1024), 0, st, d, t.primes_dev, t.L, t.log_n, rot, high); break;
        }
        PFHE_HIP(hipGetLastError());
        return PFHE_OK;
    }
    if (stream_is_capturing((hipStream_t)stream)) {
        set_last_error("mul_monomial_assign allocates a scratch tile; capture mul_monomial_to into a graph instead");
        return PFHE_ERR_UNSUPPORTED;
    }
    // a rotation cannot be done in place by independent threads: rotate tiles of up to 1 GiB into a stream-ordered
    // scratch buffer and copy them back (2x the traffic of the out-of-place form)
    hipStream_t s = (hipStream_t)stream;
    const size_t unit = t.n * t.L;
    const size_t tile = std::max<size_t>(unit, (((size_t)1 << 27) / unit) * unit);  // words
    const size_t scratch_words = std::min(tile, len);
    u64 *scratch = nullptr;
    PFHE_HIP(counted_malloc_async((void **)&scratch, scratch_words * sizeof(u64), s));
    int st = PFHE_OK;
    for (size_t off = 0; off < len && st == PFHE_OK; off += tile) {
        const size_t w = std::min(tile, len - off);
        st = monomial_to(t, (const u64 *)data_dev + off, r, scratch, w, s);
        if (st == PFHE_OK && hipMemcpyAsync(data_dev + off, scratch, w * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
            st = PFHE_ERR_HIP;
    }
    (void)counted_free_async(scratch, s);
    return st;
    PFHE_GUARD_END
}

int pfhe_dcrt_inv_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, uint64_t *out_dev, size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (int st = check_common(table, a_dev, a_dev, out_dev, len)) return st;
    DeviceGuard guard(capi_table_of(table)->device);
    if (!guard.ok) return PFHE_ERR_NO_DEVICE;
    if (len == 0) return PFHE_OK;
    const TableSet &t = *capi_table_of(table);
    hipStream_t s = (hipStream_t)stream;
    if (stream_is_capturing(s)) {
        set_last_error("inv reports non-invertible elements synchronously and cannot be captured into a graph");
        return PFHE_ERR_UNSUPPORTED;
    }
    unsigned int *flag = nullptr;
    PFHE_HIP(counted_malloc_async((void **)&flag, sizeof(unsigned int), s));
    PFHE_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned int), s));
    const dim3 th(kEwThreads);
    if (t.log_n >= 4) {
        hipLaunchKernelGGL(inv_kernel<16>, dim3(ew_grid(len / 16)), th, 0, s, (u64 *)out_dev, (const u64 *)a_dev,
                           t.primes_dev, t.L, t.log_n, len, flag);
    } else {
        hipLaunchKernelGGL(inv_kernel<1>, dim3(ew_grid(len)), th, 0, s, (u64 *)out_dev, (const u64 *)a_dev, t.primes_dev,
                           t.L, t.log_n, len, flag);
    }
    unsigned int host_flag = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&host_flag, flag, sizeof(unsigned int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)counted_free_async(flag, s);
    PFHE_HIP(e);
    if (host_flag) {
        set_last_error("an element has no inverse (zero residue)");
        return PFHE_ERR_NO_INVERSE;
    }
    return PFHE_OK;
    PFHE_GUARD_END
}

}  // extern "C"
