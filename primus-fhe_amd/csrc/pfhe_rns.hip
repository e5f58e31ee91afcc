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

// ---- unfused kernels: one reference slice function each ----
// RT: RnsDev (constants by value) or RnsWide (device table); BT: BasisDev or BasisWide.  A by-value instantiation is
// compiled for the exact limb count LEN = value_len; a wide one for value_len rounded up (zero top limbs) and addresses
// memory with the run-time value_len.  WT: the caller's word type (u64 or u32, see pfhe_rns.hpp).

template <int LEN, class RT, class WT>
__global__ __launch_bounds__(kThreads) void compose_kernel(RT R, const WT *__restrict__ multi,
                                                           WT *__restrict__ out, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    u64 v[LEN];
    if constexpr (kByValue<RT>) {
        u64 r[kMaxLimbs];
        for (u32 i = 0; i < R.L; ++i) r[i] = multi[(u64)i * count + c];
        compose<LEN>(R, r, v);
    } else {
        compose_general<LEN>(R, [&](u32 i) { return (u64)multi[(u64)i * count + c]; }, v);
    }
    const u32 vw = words_of<LEN, WT>(R);
    store_limbs<LEN, WT>(out + c * vw, vw, v);
}

template <class RT, class WT>
__global__ __launch_bounds__(kThreads) void wrapping_decompose_kernel(RT R, const WT *__restrict__ small,
                                                                     WT *__restrict__ multi, u64 count,
                                                                     u64 small_modulus) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u64 v = small[c];
    const u64 half = (small_modulus + 1) / 2;
    for (u32 i = 0; i < R.L; ++i) {
        u64 o = v;
        if (small_modulus != 2 && v >= half) o = R.modulus(i) - small_modulus + v;
        multi[(u64)i * count + c] = (WT)o;
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
template <class RT, int MAXL, class WT>
__global__ __launch_bounds__(kThreads) void add_decompose_scaled_kernel(RT R, const WT *__restrict__ small,
                                                                       WT *__restrict__ acc, u64 count,
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
        acc[idx] = (WT)add_mod(acc[idx], mul_shoup(lifted, F.value[i], F.quotient[i], q), q);
    }
}

// RNSBase::decompose_big_uint_values_to (base.rs:457-481): value mod q_i by Horner over the words, most significant
// first; every step reduces hi:lo < q * 2^64 (the words are re-read per modulus: they stay in the thread's cache lines)
template <class RT, class WT>
__global__ __launch_bounds__(kThreads) void decompose_big_kernel(RT R, u32 words, const WT *__restrict__ values,
                                                                 WT *__restrict__ multi, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const WT *__restrict__ v = values + c * words;
    for (u32 j = 0; j < R.L; ++j) {
        const u64 p = R.modulus(j), lo = R.ratio_lo(j), hi = R.ratio_hi(j);
        u64 r = 0;
        if constexpr (sizeof(WT) == 8) {
            for (u32 k = words; k-- > 0;) r = barrett_reduce128(v[k], r, p, lo, hi);
        } else {  // r < q < 2^30: r * 2^32 + word < 2^62
            for (u32 k = words; k-- > 0;) r = barrett_reduce128((r << 32) | v[k], 0, p, lo, hi);
        }
        multi[(u64)j * count + c] = (WT)r;
    }
}

template <int LEN, class BT, class WT>
__global__ __launch_bounds__(kThreads) void init_value_carry_kernel(BT B, WT *__restrict__ values,
                                                                    unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 vw = words_of<LEN, WT>(B);
    u64 v[LEN];
    load_limbs<LEN, WT>(values + c * vw, vw, v);
    const u32 carry = init_value_carry<LEN>(B, v);
    store_limbs<LEN, WT>(values + c * vw, vw, v);
    carries[c] = (unsigned char)carry;
}

template <class WT>
__global__ __launch_bounds__(kThreads) void unsigned_decompose_kernel(BasisCore B, u32 level,
                                                                     const WT *__restrict__ values,
                                                                     WT *__restrict__ digits,
                                                                     unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 start = B.drop_bits + level * B.log_basis;
    const u64 temp = window_words<WT>(values + c * B.value_words, start, B.basis_minus_one, B.log_basis) + carries[c];
    carries[c] = (temp & B.carry_mask) != 0;  // common.rs:275-285
    digits[c] = (WT)(temp & B.basis_minus_one);
}

// common.rs:255-272 over a slice (:289-306): the signed digit as a residue modulo Q.  With the carry set the digit
// temp stands for temp - B and is stored as (Q - B) + temp; temp == B is the digit 0.
template <class RT, class WT>
__global__ __launch_bounds__(kThreads) void signed_decompose_kernel(RT R, BasisCore B, u32 level,
                                                                   const WT *__restrict__ values,
                                                                   WT *__restrict__ out,
                                                                   unsigned char *__restrict__ carries, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const u32 len = B.value_words;
    const u32 start = B.drop_bits + level * B.log_basis;
    const u64 temp = window_words<WT>(values + c * len, start, B.basis_minus_one, B.log_basis) + carries[c];
    const bool carry = (temp & B.carry_mask) != 0;
    carries[c] = carry;
    WT *d = out + c * len;
    if (carry && temp <= B.basis_minus_one) {
        // Q - (B - temp), word by word with borrow (B - temp >= 1)
        u64 sub = B.basis - temp;
        for (u32 j = 0; j < len; ++j) {
            const u64 q = product_word<WT>(R, j);
            d[j] = (WT)(q - sub);
            sub = q < sub ? 1 : 0;
        }
    } else {
        for (u32 j = 0; j < len; ++j) d[j] = 0;
        if (!carry) d[0] = (WT)temp;
    }
}

// ---- fused steps (1)-(4): one thread per coefficient, big integer kept in registers ----
template <int LEN, class RT, class BT, class WT>
__global__ __launch_bounds__(kThreads) void gadget_decompose_kernel(RT R, BT B, u32 log_n,
                                                                   const WT *__restrict__ crt, WT *__restrict__ out,
                                                                   u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 n = 1ull << log_n;
    const u64 poly = gid >> log_n, t = gid & (n - 1);
    u64 v[LEN];
    if (R.big_input) {  // glwe/dcrt.rs:258-338: the polynomial arrives composed
        const u32 vw = words_of<LEN, WT>(R);
        load_limbs<LEN, WT>(crt + (poly * n + t) * vw, vw, v);
    } else {
        const WT *__restrict__ in = crt + poly * R.L * n + t;
        if constexpr (kByValue<RT>) {
            u64 r[kMaxLimbs];
            for (u32 i = 0; i < R.L; ++i) r[i] = in[(u64)i * n];
            compose<LEN>(R, r, v);
        } else {
            compose_general<LEN>(R, [&](u32 i) { return (u64)in[(u64)i * n]; }, v);
        }
    }
    u32 carry = init_value_carry<LEN>(B, v);
    const u64 half = (B.basis + 1) / 2;
    WT *__restrict__ o = out + poly * B.ell * R.L * n + t;
    for (u32 j = 0; j < B.ell; ++j) {
        const u64 temp = window<LEN>(v, B.drop_bits + j * B.log_basis, B.basis_minus_one, B.log_basis) + carry;
        carry = (temp & B.carry_mask) != 0;
        const u64 u = temp & B.basis_minus_one;
        for (u32 i = 0; i < R.L; ++i) {
            u64 res = u;
            if (B.basis != 2 && u >= half) res = R.modulus(i) - B.basis + u;  // centred lift, base.rs:721-730
            o[((u64)j * R.L + i) * n] = (WT)res;
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

// the same sum for a U32DcrtTable (dcrt/prime32.rs:11): residues below 2^30, products below 2^60, eight of them (and a
// canonical accumulator) below 2^63 + 2^30 — folded every 8 terms with one Barrett step on a single word
// (bar = floor(2^64 / q): the quotient estimate is exact or one short for any 64-bit input)
__device__ __forceinline__ u32 red64_32(u64 x, u32 q, u64 bar) {
    const u64 r = x - mulhi64(x, bar) * q;
    return (u32)(r >= q ? r - q : r);
}
__global__ __launch_bounds__(kThreads) void gadget_mulacc32_kernel(const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                                   u32 k, u32 rows, u32 ell, const u32 *__restrict__ digits,
                                                                   const u32 *__restrict__ ggsw, u64 ggsw_stride,
                                                                   u32 *__restrict__ result, u64 total, u32 accumulate) {
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
    const u32 q = (u32)primes[limb].q;
    const u64 bar = primes[limb].bar_lo;
    const u32 *__restrict__ dg = digits + e * rows * ell * W + (u64)limb * n + t;
    const u32 *__restrict__ key = ggsw + e * ggsw_stride + ((u64)c * L + limb) * n + t;
    u32 *__restrict__ out = result + (e * (k + 1) + c) * W + (u64)limb * n + t;
    u64 acc = accumulate ? *out : 0u;
    const u32 terms = rows * ell;
    u32 pending = accumulate ? 1u : 0u;  // the previous result counts as one term (below 2^32 <= 2^60)
    for (u32 ij = 0; ij < terms; ++ij) {
        acc += (u64)dg[(u64)ij * W] * key[(u64)ij * (k + 1) * W];
        if (++pending == kFold32Every) {
            acc = red64_32(acc, q, bar);
            pending = 0;
        }
    }
    *out = red64_32(acc, q, bar);
}

template <class WT>
struct Launch {
    template <int LEN>
    struct Compose {
        static int run(const RnsParams &r, const WT *multi, WT *out, u64 count, hipStream_t s) {
            const dim3 g(grid_for(count)), th(kThreads);
            if (r.wide()) hipLaunchKernelGGL((compose_kernel<LEN, RnsWide, WT>), g, th, 0, s, r.wide_tab, multi, out, count);
            else if constexpr (LEN <= kMaxLimbs) hipLaunchKernelGGL((compose_kernel<LEN, RnsDev, WT>), g, th, 0, s, r.dev, multi, out, count);
            return PFHE_OK;
        }
    };
    template <int LEN>
    struct Init {
        static int run(const BasisParams &b, WT *values, unsigned char *carries, u64 count, hipStream_t s) {
            const dim3 g(grid_for(count)), th(kThreads);
            if (b.wide()) hipLaunchKernelGGL((init_value_carry_kernel<LEN, BasisWide, WT>), g, th, 0, s, b.wide_tab, values, carries, count);
            else if constexpr (LEN <= kMaxLimbs) hipLaunchKernelGGL((init_value_carry_kernel<LEN, BasisDev, WT>), g, th, 0, s, b.dev, values, carries, count);
            return PFHE_OK;
        }
    };
    template <int LEN>
    struct Fused {
        static int run(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt, WT *out, u64 total, hipStream_t s) {
            const dim3 g(grid_for(total)), th(kThreads);
            if (b.wide()) {  // value_len <= L: a wide basis implies a wide base
                hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsWide, BasisWide, WT>), g, th, 0, s, r.wide_tab, b.wide_tab, log_n, crt, out, total);
            } else if constexpr (LEN <= kMaxLimbs) {
                if (r.wide()) hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsWide, BasisDev, WT>), g, th, 0, s, r.wide_tab, b.dev, log_n, crt, out, total);
                else hipLaunchKernelGGL((gadget_decompose_kernel<LEN, RnsDev, BasisDev, WT>), g, th, 0, s, r.dev, b.dev, log_n, crt, out, total);
            }
            return PFHE_OK;
        }
    };
};

// the moduli and their Barrett ratios by value, for decompose_big_kernel on a base of at most kMaxLimbs moduli
struct ModuliByValue {
    u32 L, pad_;
    u64 q[kMaxLimbs], lo[kMaxLimbs], hi[kMaxLimbs];
    __device__ u64 modulus(u32 i) const { return q[i]; }
    __device__ u64 ratio_lo(u32 i) const { return lo[i]; }
    __device__ u64 ratio_hi(u32 i) const { return hi[i]; }
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
    p.wide_tab = RnsWide{p.dev.L, p.dev.value_len, 0u, p.dev.value_words, (const u64 *)p.blob->ptr};
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

template <class WT>
int rns_compose_dev(const RnsParams &r, const WT *multi, WT *out, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<Launch<WT>::template Compose>(r.dev.value_len, r, multi, out, count, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int rns_wrapping_decompose_dev(const RnsParams &r, const WT *small, WT *multi, u64 count, u64 small_modulus,
                               hipStream_t s) {
    if (count == 0) return PFHE_OK;
    const dim3 g(grid_for(count)), th(kThreads);
    if (r.wide()) hipLaunchKernelGGL((wrapping_decompose_kernel<RnsWide, WT>), g, th, 0, s, r.wide_tab, small, multi, count, small_modulus);
    else hipLaunchKernelGGL((wrapping_decompose_kernel<RnsDev, WT>), g, th, 0, s, r.dev, small, multi, count, small_modulus);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class RT, int MAXL, class WT>
static void launch_add_scaled(const RT &R, const WT *small_values, WT *acc, u64 value_count, u64 small_value_modulus,
                              bool centred, const u64 *factor_pairs, hipStream_t s) {
    ScaledFactors<MAXL> f{};
    for (u32 i = 0; i < R.L; ++i) {
        f.value[i] = factor_pairs[2 * i];
        f.quotient[i] = factor_pairs[2 * i + 1];
    }
    hipLaunchKernelGGL((add_decompose_scaled_kernel<RT, MAXL, WT>), dim3(grid_for(value_count)), dim3(kThreads), 0, s, R,
                       small_values, acc, value_count, small_value_modulus, centred, f);
}

template <class WT>
int rns_add_decompose_scaled_dev(const RnsParams &r, const WT *small_values, WT *acc, u64 value_count,
                                 u64 small_value_modulus, bool centred, const u64 *factor_pairs, hipStream_t s) {
    if (value_count == 0) return PFHE_OK;
    if (r.wide()) launch_add_scaled<RnsWide, kMaxWideLimbs, WT>(r.wide_tab, small_values, acc, value_count, small_value_modulus, centred, factor_pairs, s);
    else launch_add_scaled<RnsDev, kMaxLimbs, WT>(r.dev, small_values, acc, value_count, small_value_modulus, centred, factor_pairs, s);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int rns_decompose_big_dev(const RnsParams &r, const WT *values, WT *multi, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    const dim3 g(grid_for(count)), th(kThreads);
    const u32 words = sizeof(WT) == 8 ? r.dev.value_len : r.dev.value_words;
    if (r.wide()) {
        hipLaunchKernelGGL((decompose_big_kernel<RnsWide, WT>), g, th, 0, s, r.wide_tab, words, values, multi, count);
    } else {
        ModuliByValue m{};
        m.L = r.dev.L;
        for (u32 i = 0; i < m.L; ++i) {
            m.q[i] = r.dev.q[i];
            const unsigned __int128 top = (unsigned __int128)1 << 64;  // floor(2^128 / q), high word then low
            m.hi[i] = (u64)(top / m.q[i]);
            m.lo[i] = (u64)(((top % m.q[i]) << 64) / m.q[i]);
        }
        hipLaunchKernelGGL((decompose_big_kernel<ModuliByValue, WT>), g, th, 0, s, m, words, values, multi, count);
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int basis_init_value_carry_dev(const BasisParams &b, WT *values, unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<Launch<WT>::template Init>(b.dev.value_len, b, values, carries, count, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int basis_unsigned_decompose_dev(const BasisParams &b, u32 level, const WT *values, WT *digits,
                                 unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    hipLaunchKernelGGL(unsigned_decompose_kernel<WT>, dim3(grid_for(count)), dim3(kThreads), 0, s,
                       static_cast<const BasisCore &>(b.dev), level, values, digits, carries, count);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int basis_signed_decompose_dev(const RnsParams &r, const BasisParams &b, u32 level, const WT *values, WT *decomposed,
                               unsigned char *carries, u64 count, hipStream_t s) {
    if (count == 0) return PFHE_OK;
    const dim3 g(grid_for(count)), th(kThreads);
    const BasisCore &core = b.dev;
    if (r.wide()) hipLaunchKernelGGL((signed_decompose_kernel<RnsWide, WT>), g, th, 0, s, r.wide_tab, core, level, values, decomposed, carries, count);
    else hipLaunchKernelGGL((signed_decompose_kernel<RnsDev, WT>), g, th, 0, s, r.dev, core, level, values, decomposed, carries, count);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class WT>
int gadget_decompose_dev(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt, WT *digits, u64 npolys,
                         hipStream_t s) {
    const u64 total = npolys << log_n;
    if (total == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<Launch<WT>::template Fused>(r.dev.value_len, r, b, log_n, crt, digits, total, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// the two word types of the C ABI
#define PFHE_INSTANTIATE(WT)                                                                                               \
    template int rns_compose_dev<WT>(const RnsParams &, const WT *, WT *, u64, hipStream_t);                               \
    template int rns_wrapping_decompose_dev<WT>(const RnsParams &, const WT *, WT *, u64, u64, hipStream_t);               \
    template int rns_add_decompose_scaled_dev<WT>(const RnsParams &, const WT *, WT *, u64, u64, bool, const u64 *, hipStream_t); \
    template int rns_decompose_big_dev<WT>(const RnsParams &, const WT *, WT *, u64, hipStream_t);                         \
    template int basis_init_value_carry_dev<WT>(const BasisParams &, WT *, unsigned char *, u64, hipStream_t);             \
    template int basis_unsigned_decompose_dev<WT>(const BasisParams &, u32, const WT *, WT *, unsigned char *, u64, hipStream_t); \
    template int basis_signed_decompose_dev<WT>(const RnsParams &, const BasisParams &, u32, const WT *, WT *, unsigned char *, u64, hipStream_t); \
    template int gadget_decompose_dev<WT>(const RnsParams &, const BasisParams &, u32, const WT *, WT *, u64, hipStream_t);
PFHE_INSTANTIATE(u64)
PFHE_INSTANTIATE(u32)
#undef PFHE_INSTANTIATE

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

int gadget_mulacc32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 k, u32 rows, u32 ell, const u32 *digits,
                        const u32 *ggsw, bool ggsw_shared, u32 *result, u64 batch, bool accumulate, hipStream_t s) {
    const u64 total = (batch * (k + 1) * L) << log_n;
    if (total == 0) return PFHE_OK;
    const u64 ggsw_words = ((u64)rows * ell * (k + 1) * L) << log_n;
    hipLaunchKernelGGL(gadget_mulacc32_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, primes, L, log_n, k, rows, ell,
                       digits, ggsw, ggsw_shared ? 0ull : ggsw_words, result, total, accumulate ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe
