// pfhe_rns.hip — RNS composition, gadget decomposition and the digit x key multiply-accumulate.
//
// Reference steps (primus_lattice/src/glwe/dcrt.rs:178-255), per input CRT polynomial:
//   (1) RNSBase::compose_multiple_values_to          primus_rns/src/base.rs:648-675 (-> :609-633)
//   (2) BigUintApproxSignedBasis::init_value_carry_slice_inplace
//                                                     primus_decompose/src/big_integer/basis.rs:326-367
//   (3) OnceBigUintSignedDecomposer::unsigned_decompose_slice_to   big_integer/common.rs:275-325
//   (4) RNSBase::wrapping_decompose_small_values_to   primus_rns/src/base.rs:279-312,721-730
//   (5) DcrtTable::transform_slice                    (pfhe_ntt.hip)
//   (6) DcrtGlwe::add_dcrt_glwe_mul_dcrt_polynomial_assign         glwe/dcrt.rs:108-126
// Each step exists as its own kernel (parity with the reference's slice functions), and steps
// (1)-(4) additionally as ONE fused kernel that keeps the composed big integer in registers and
// never writes it (the reference writes N*value_len words of scratch and re-reads them ell times).
// All values are exact integers; every output is canonical.
#include <type_traits>

#include "pfhe_modmath.hpp"
#include "pfhe_rns.hpp"
#include "pfhe_rns_device.hpp"
#include "pfhe_staging.hpp"

namespace pfhe {

namespace {

constexpr int kThreads = 256;

u32 grid_for(u64 items) {
    u64 g = (items + kThreads - 1) / kThreads;
    if (g == 0) g = 1;
    if (g > 0x7fffffffull) g = 0x7fffffffull;
    return (u32)g;
}

__device__ __forceinline__ u64 window_dyn(const u64 *v, u32 start, u64 mask, u32 log_basis) {
    const u32 idx = start >> 6, shr = start & 63;
    u64 w = v[idx] >> shr;
    if (shr + log_basis > 64) w |= v[idx + 1] << (64 - shr);
    return w & mask;
}

// ---- unfused kernels: one reference slice function each ----
// RT: RnsDev (constants by value) or RnsWide (device table); BT: BasisDev or BasisWide.  A by-value instantiation is
// compiled for the exact limb count LEN = value_len; a wide one for value_len rounded up (zero top limbs) and addresses
// memory with the run-time value_len.

template <int LEN, class RT>
__global__ __launch_bounds__(kThreads) void compose_kernel(RT R, const u64 *__restrict__ multi,
                                                           u64 *__restrict__ out, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    u64 v[LEN];
    if constexpr (kByValue<RT>) {
        u64 r[kMaxLimbs];
        for (u32 i = 0; i < R.L; ++i) r[i] = multi[(u64)i * count + c];
        compose<LEN>(R, r, v);
    } else {
        compose_general<LEN>(R, [&](u32 i) { return multi[(u64)i * count + c]; }, v);
    }
    const u32 vl = kByValue<RT> ? (u32)LEN : R.value_len;
#pragma unroll
    for (int j = 0; j < LEN; ++j)
        if ((u32)j < vl) out[c * vl + j] = v[j];
}

template <class RT>
__global__ __launch_bounds__(kThreads) void wrapping_decompose_kernel(RT R, const u64 *__restrict__ small,
                                                                     u64 *__restrict__ multi, u64 count,
                                                                     u64 small_modulus) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u64 v = small[c];
    const u64 half = (small_modulus + 1) / 2;
    for (u32 i = 0; i < R.L; ++i) {
        u64 o = v;
        if (small_modulus != 2 && v >= half) o = R.modulus(i) - small_modulus + v;
        multi[(u64)i * count + c] = o;
    }
}

// acc[i][c] = reduce_add(acc[i][c], factor_i * lift_i(small[c])): RNSBase::add_wrapping_decompose_small_values_scaled
// (base.rs:326-384, slice::wrapping_decompose_chunk_scaled_to :739-757) when `centred`, else
// add_decompose_small_values_scaled (base.rs:398-416; also the reference's small_value_modulus == 2 branch).
// `fv` / `fq`: the ShoupFactor (value, quotient) of each modulus.
template <int MAXL>
struct ScaledFactors {
    u64 value[MAXL], quotient[MAXL];
};
template <class RT, int MAXL>
__global__ __launch_bounds__(kThreads) void add_decompose_scaled_kernel(RT R, const u64 *__restrict__ small,
                                                                       u64 *__restrict__ acc, u64 count,
                                                                       u64 small_modulus, bool centred,
                                                                       ScaledFactors<MAXL> F) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u64 v = small[c];
    const u64 half = (small_modulus + 1) / 2;
    for (u32 i = 0; i < R.L; ++i) {
        const u64 q = R.modulus(i);
        const u64 lifted = (centred && v >= half) ? q - small_modulus + v : v;
        const u64 idx = (u64)i * count + c;
        acc[idx] = add_mod(acc[idx], mul_shoup(lifted, F.value[i], F.quotient[i], q), q);
    }
}

template <int LEN, class BT>
__global__ __launch_bounds__(kThreads) void init_value_carry_kernel(BT B, u64 *__restrict__ values,
                                                                    unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 vl = kByValue<BT> ? (u32)LEN : B.value_len;
    u64 v[LEN];
#pragma unroll
    for (int j = 0; j < LEN; ++j) v[j] = (u32)j < vl ? values[c * vl + j] : 0;
    const u32 carry = init_value_carry<LEN>(B, v);
#pragma unroll
    for (int j = 0; j < LEN; ++j)
        if ((u32)j < vl) values[c * vl + j] = v[j];
    carries[c] = (unsigned char)carry;
}

__global__ __launch_bounds__(kThreads) void unsigned_decompose_kernel(BasisCore B, u32 level,
                                                                     const u64 *__restrict__ values,
                                                                     u64 *__restrict__ digits,
                                                                     unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 start = B.drop_bits + level * B.log_basis;
    const u64 temp = window_dyn(values + c * B.value_len, start, B.basis_minus_one, B.log_basis) + carries[c];
    carries[c] = (temp & B.carry_mask) != 0;  // common.rs:275-285
    digits[c] = temp & B.basis_minus_one;
}

// common.rs:255-272 over a slice (:289-306): the signed digit as a residue modulo Q.  With the carry set the digit
// temp stands for temp - B and is stored as (Q - B) + temp; temp == B is the digit 0.
template <class RT>
__global__ __launch_bounds__(kThreads) void signed_decompose_kernel(RT R, BasisCore B, u32 level,
                                                                   const u64 *__restrict__ values,
                                                                   u64 *__restrict__ out,
                                                                   unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 len = B.value_len;
    const u32 start = B.drop_bits + level * B.log_basis;
    const u64 temp = window_dyn(values + c * len, start, B.basis_minus_one, B.log_basis) + carries[c];
    const bool carry = (temp & B.carry_mask) != 0;
    carries[c] = carry;
    u64 *d = out + c * len;
    if (carry && temp <= B.basis_minus_one) {
        // Q - (B - temp), limb by limb with borrow (B - temp >= 1)
        u64 sub = B.basis - temp;
        for (u32 j = 0; j < len; ++j) {
            const u64 q = R.product(j);
            d[j] = q - sub;
            sub = q < sub ? 1 : 0;
        }
    } else {
        for (u32 j = 0; j < len; ++j) d[j] = 0;
        if (!carry) d[0] = temp;
    }
}

// ---- fused steps (1)-(4): one thread per coefficient, big integer kept in registers ----
template <int LEN, class RT, class BT>
__global__ __launch_bounds__(kThreads) void gadget_decompose_kernel(RT R, BT B, u32 log_n,
                                                                   const u64 *__restrict__ crt, u64 *__restrict__ out,
                                                                   u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 n = 1ull << log_n;
    const u64 poly = gid >> log_n, t = gid & (n - 1);
    const u32 vl = kByValue<RT> ? (u32)LEN : R.value_len;
    u64 v[LEN];
    if (R.big_input) {  // glwe/dcrt.rs:258-338: the polynomial arrives composed
#pragma unroll
        for (int j = 0; j < LEN; ++j) v[j] = (u32)j < vl ? crt[(poly * n + t) * vl + j] : 0;
    } else {
        const u64 *__restrict__ in = crt + poly * R.L * n + t;
        if constexpr (kByValue<RT>) {
            u64 r[kMaxLimbs];
            for (u32 i = 0; i < R.L; ++i) r[i] = in[(u64)i * n];
            compose<LEN>(R, r, v);
        } else {
            compose_general<LEN>(R, [&](u32 i) { return in[(u64)i * n]; }, v);
        }
    }
    u32 carry = init_value_carry<LEN>(B, v);
    const u64 half = (B.basis + 1) / 2;
    u64 *__restrict__ o = out + poly * B.ell * R.L * n + t;
    for (u32 j = 0; j < B.ell; ++j) {
        const u64 temp = window<LEN>(v, B.drop_bits + j * B.log_basis, B.basis_minus_one, B.log_basis) + carry;
        carry = (temp & B.carry_mask) != 0;
        const u64 u = temp & B.basis_minus_one;
        for (u32 i = 0; i < R.L; ++i) {
            u64 res = u;
            if (B.basis != 2 && u >= half) res = R.modulus(i) - B.basis + u;  // centred lift, base.rs:721-730
            o[((u64)j * R.L + i) * n] = res;
        }
    }
}

// ---- step (6) summed over rows and levels with ONE Barrett reduction per output word ----
// Products of canonical residues (< 2^124) are summed lazily in 128 bits and folded every 8 terms,
// so the final canonical value equals what the reference reaches with rows*ell sequential
// reduce_mul_add calls (exact integer arithmetic; the order of reductions cannot change it).
__global__ __launch_bounds__(kThreads) void gadget_mulacc_kernel(const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                                 u32 k, u32 rows, u32 ell, const u64 *__restrict__ digits,
                                                                 const u64 *__restrict__ ggsw, u64 ggsw_stride,
                                                                 u64 *__restrict__ result, u64 total, u32 accumulate) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 n = 1ull << log_n;
    const u64 W = (u64)L * n;
    const u64 t = gid & (n - 1);
    u64 rest = gid >> log_n;
    const u32 limb = (u32)(rest % L);
    rest /= L;
    const u32 c = (u32)(rest % (k + 1));
    const u64 e = rest / (k + 1);
    const NttPrime *P = primes + limb;
    const u64 *__restrict__ dg = digits + e * rows * ell * W + (u64)limb * n + t;
    const u64 *__restrict__ key = ggsw + e * ggsw_stride + ((u64)c * L + limb) * n + t;
    u64 lo = 0, hi = 0;
    const u32 terms = rows * ell;
    for (u32 ij = 0; ij < terms; ++ij) {
        const u64 d = dg[(u64)ij * W];
        const u64 g = key[(u64)ij * (k + 1) * W];
        const u64 pl = d * g, ph = mulhi64(d, g);
        lo += pl;
        hi += ph + (lo < pl);
        if ((ij & 7u) == 7u) {  // 8 products of residues < 2^62 stay below 2^127: fold before the next 8
            lo = barrett_reduce128(lo, hi, P->q, P->bar_lo, P->bar_hi);
            hi = 0;
        }
    }
    u64 *__restrict__ out = result + (e * (k + 1) + c) * W + (u64)limb * n + t;
    if (accumulate) {
        const u64 a = *out;
        lo += a;
        hi += lo < a;
    }
    *out = barrett_reduce128(lo, hi, P->q, P->bar_lo, P->bar_hi);
}

template <int LEN>
struct ComposeLaunch {
    static int run(const RnsParams &r, const u64 *multi, u64 *out, u64 count, hipStream_t s) {
        const dim3 g(grid_for(count)), th(kThreads);
        if (r.wide()) hipLaunchKernelGGL((compose_kernel<LEN, RnsWide>), g, th, 0, s, r.wide_tab, multi, out, count);
        else if constexpr (LEN <= kMaxLimbs) hipLaunchKernelGGL((compose_kernel<LEN, RnsDev>), g, th, 0, s, r.dev, multi, out, count);
        return PFHE_OK;
    }
};
template <int LEN>
struct InitLaunch {
    static int run(const BasisParams &b, u64 *values, unsigned char *carries, u64 count, hipStream_t s) {
        const dim3 g(grid_for(count)), th(kThreads);
        if (b.wide()) hipLaunchKernelGGL((init_value_carry_kernel<LEN, BasisWide>), g, th, 0, s, b.wide_tab, values, carries, count);
        else if constexpr (LEN <= kMaxLimbs) hipLaunchKernelGGL((init_value_carry_kernel<LEN, BasisDev>), g, th, 0, s, b.dev, values, carries, count);
        return PFHE_OK;
    }
};
template <int LEN>
struct FusedLaunch {
    static int run(const RnsParams &r, const BasisParams &b, u32 log_n, const u64 *crt, u64 *out, u64 total, hipStream_t s) {
        const dim3 g(grid_for(total)), th(kThreads);
        if (b.wide()) {  // value_len <= L: a wide basis implies a wide base
            hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsWide, BasisWide>), g, th, 0, s, r.wide_tab, b.wide_tab, log_n, crt, out, total);
        } else if constexpr (LEN <= kMaxLimbs) {
            if (r.wide()) hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsWide, BasisDev>), g, th, 0, s, r.wide_tab, b.dev, log_n, crt, out, total);
            else hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsDev, BasisDev>), g, th, 0, s, r.dev, b.dev, log_n, crt, out, total);
        }
        return PFHE_OK;
    }
};

}  // namespace

DeviceBlob::~DeviceBlob() {
    if (!ptr) return;
    DeviceGuard g(device);
    (void)counted_free(ptr);
}

static int upload_table(int device, const std::vector<u64> &host, std::shared_ptr<DeviceBlob> &blob) {
    auto b = std::make_shared<DeviceBlob>();
    b->device = device;
    PFHE_HIP(counted_malloc(&b->ptr, host.size() * sizeof(u64)));
    PFHE_HIP(hipMemcpy(b->ptr, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
    blob = std::move(b);
    return PFHE_OK;
}

int upload_rns_wide(RnsHost &r) {
    RnsParams &p = r.par;
    if (!p.wide()) return PFHE_OK;
    constexpr size_t W = kMaxWideLimbs;
    const size_t L = p.dev.L, len = p.dev.value_len;
    std::vector<u64> t(RnsWide::table_words(), 0);
    for (size_t i = 0; i < L; ++i) {
        t[i] = r.moduli[i];
        t[W + i] = r.inv_punct[i];
        t[2 * W + i] = r.inv_punct_p[i];
        t[4 * W + i] = r.ratio_lo[i];
        t[5 * W + i] = r.ratio_hi[i];
        for (size_t j = 0; j < len; ++j) t[(6 + i) * W + j] = r.punct[i * len + j];
    }
    for (size_t j = 0; j < len; ++j) t[3 * W + j] = r.Q[j];
    PFHE_TRY(upload_table(r.device, t, p.blob));
    p.wide_tab = RnsWide{p.dev.L, p.dev.value_len, 0u, 0u, (const u64 *)p.blob->ptr};
    return PFHE_OK;
}

int upload_basis_wide(BasisHost &b) {
    BasisParams &p = b.par;
    if (!p.wide()) return PFHE_OK;
    constexpr size_t W = kMaxWideLimbs;
    std::vector<u64> t(2 * W, 0);
    for (size_t j = 0; j < p.dev.value_len; ++j) {
        t[j] = b.threshold[j];
        t[W + j] = b.add[j];
    }
    PFHE_TRY(upload_table(b.device, t, p.blob));
    static_cast<BasisCore &>(p.wide_tab) = static_cast<const BasisCore &>(p.dev);
    p.wide_tab.tab = (const u64 *)p.blob->ptr;
    return PFHE_OK;
}

int rns_compose_dev(const RnsParams &r, const u64 *multi, u64 *out, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<ComposeLaunch>(r.dev.value_len, r, multi, out, count, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int rns_wrapping_decompose_dev(const RnsParams &r, const u64 *small, u64 *multi, u64 count, u64 small_modulus,
                               hipStream_t s) {
    if (count == 0) return PFHE_OK;
    const dim3 g(grid_for(count)), th(kThreads);
    if (r.wide()) hipLaunchKernelGGL(wrapping_decompose_kernel<RnsWide>, g, th, 0, s, r.wide_tab, small, multi, count, small_modulus);
    else hipLaunchKernelGGL(wrapping_decompose_kernel<RnsDev>, g, th, 0, s, r.dev, small, multi, count, small_modulus);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class RT, int MAXL>
static void launch_add_scaled(const RT &R, const u64 *small_values, u64 *acc, u64 value_count, u64 small_value_modulus,
                              bool centred, const u64 *factor_pairs, hipStream_t s) {
    ScaledFactors<MAXL> f{};
    for (u32 i = 0; i < R.L; ++i) {
        f.value[i] = factor_pairs[2 * i];
        f.quotient[i] = factor_pairs[2 * i + 1];
    }
    hipLaunchKernelGGL((add_decompose_scaled_kernel<RT, MAXL>), dim3(grid_for(value_count)), dim3(kThreads), 0, s, R,
                       small_values, acc, value_count, small_value_modulus, centred, f);
}

int rns_add_decompose_scaled_dev(const RnsParams &r, const u64 *small_values, u64 *acc, u64 value_count,
                                 u64 small_value_modulus, bool centred, const u64 *factor_pairs, hipStream_t s) {
    if (value_count == 0) return PFHE_OK;
    if (r.wide()) launch_add_scaled<RnsWide, kMaxWideLimbs>(r.wide_tab, small_values, acc, value_count, small_value_modulus, centred, factor_pairs, s);
    else launch_add_scaled<RnsDev, kMaxLimbs>(r.dev, small_values, acc, value_count, small_value_modulus, centred, factor_pairs, s);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int basis_init_value_carry_dev(const BasisParams &b, u64 *values, unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<InitLaunch>(b.dev.value_len, b, values, carries, count, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int basis_unsigned_decompose_dev(const BasisParams &b, u32 level, const u64 *values, u64 *digits,
                                 unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    hipLaunchKernelGGL(unsigned_decompose_kernel, dim3(grid_for(count)), dim3(kThreads), 0, s,
                       static_cast<const BasisCore &>(b.dev), level, values, digits, carries, count);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int basis_signed_decompose_dev(const RnsParams &r, const BasisParams &b, u32 level, const u64 *values, u64 *decomposed,
                               unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    const dim3 g(grid_for(count)), th(kThreads);
    const BasisCore &core = b.dev;
    if (r.wide()) hipLaunchKernelGGL(signed_decompose_kernel<RnsWide>, g, th, 0, s, r.wide_tab, core, level, values, decomposed, carries, count);
    else hipLaunchKernelGGL(signed_decompose_kernel<RnsDev>, g, th, 0, s, r.dev, core, level, values, decomposed, carries, count);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int gadget_decompose_dev(const RnsParams &r, const BasisParams &b, u32 log_n, const u64 *crt, u64 *digits, u64 npolys,
                         hipStream_t s) {
    const u64 total = npolys << log_n;
    if (total == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<FusedLaunch>(r.dev.value_len, r, b, log_n, crt, digits, total, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int gadget_mulacc_dev(const NttPrime *primes, u32 L, u32 log_n, u32 k, u32 rows, u32 ell, const u64 *digits,
                      const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate, hipStream_t s) {
    const u64 total = (batch * (k + 1) * L) << log_n;
    if (total == 0) return PFHE_OK;
    const u64 ggsw_words = ((u64)rows * ell * (k + 1) * L) << log_n;
    hipLaunchKernelGGL(gadget_mulacc_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, primes, L, log_n, k, rows, ell,
                       digits, ggsw, ggsw_shared ? 0ull : ggsw_words, result, total, accumulate ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe
