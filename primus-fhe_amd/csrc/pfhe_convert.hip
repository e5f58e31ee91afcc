// pfhe_convert.hip — RNS base conversion and big-integer decomposition on the GPU.
//
//   BaseConverter::fast_convert_array   primus_rns/src/converter.rs:192-218 (scratch fill :144-178)
//   BaseConverter::exact_convert_array  primus_rns/src/converter.rs:274-364
//   RNSBase::decompose_big_uint_values_to   primus_rns/src/base.rs:457-481
//
// One thread owns one coefficient: it reads its L_in residues (modulus-major input, so a wave reads
// L_in contiguous 512-byte rows), scales each by (Q/q_i)^-1 mod q_i, and produces every output
// residue as a modular dot product against the row (Q/q_i) mod p_j.  The reference's coefficient-
// major scratch array never exists: the scaled residues stay in registers.  Arithmetic follows the
// reference step by step (128-bit accumulation of the dot product, one BarrettModulus::reduce,
// barrett/mod.rs:99-139), so results are identical, not merely congruent.
// exact_convert_array is the only floating-point code near this path: f64 quotients temp_i / q_i
// (IEEE division, correctly rounded on gfx950 as in Rust), summed left to right, (sum + 0.5) as u64.
#include <memory>
#include <vector>

#include "pfhe_capi_internal.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_rns.hpp"
#include "pfhe_staging.hpp"

namespace pfhe {

// Constants by value (both bases of at most kMaxLimbs moduli) ...
struct ConvDev {
    u32 lin, lout;
    u64 q[kMaxLimbs], inv[kMaxLimbs], inv_p[kMaxLimbs];       // input base
    u64 p[kMaxLimbs], mu_lo[kMaxLimbs], mu_hi[kMaxLimbs];     // output base + floor(2^128/p)
    u64 m[kMaxLimbs][kMaxLimbs];                              // m[j][i] = (Q/q_i) mod p_j
    u64 q_mod_p0;                                             // Q mod p_0
    static constexpr int kScaled = kMaxLimbs;                 // per-thread scaled residues
    __device__ u64 q_in(u32 i) const { return q[i]; }
    __device__ u64 inv_in(u32 i) const { return inv[i]; }
    __device__ u64 inv_in_p(u32 i) const { return inv_p[i]; }
    __device__ u64 p_out(u32 j) const { return p[j]; }
    __device__ u64 ratio_lo(u32 j) const { return mu_lo[j]; }
    __device__ u64 ratio_hi(u32 j) const { return mu_hi[j]; }
    __device__ u64 matrix(u32 j, u32 i) const { return m[j][i]; }
};
// ... or, when either base is wider (BaseConverter::new takes any two bases, converter.rs:43-69), in a device table:
// q[W] | inv[W] | inv_p[W] | p[W] | mu_lo[W] | mu_hi[W] | m[W][W], W = kMaxWideLimbs.  SCALED: the per-thread array of
// scaled residues, lin rounded up to a multiple of 8 (its loops are unrolled over SCALED so that it stays in registers).
template <int SCALED>
struct ConvWide {
    u32 lin, lout;
    const u64 *tab;
    u64 q_mod_p0;
    static constexpr int kScaled = SCALED;
    static constexpr u32 W = kMaxWideLimbs;
    __device__ u64 q_in(u32 i) const { return tab[i]; }
    __device__ u64 inv_in(u32 i) const { return tab[W + i]; }
    __device__ u64 inv_in_p(u32 i) const { return tab[2 * W + i]; }
    __device__ u64 p_out(u32 j) const { return tab[3 * W + j]; }
    __device__ u64 ratio_lo(u32 j) const { return tab[4 * W + j]; }
    __device__ u64 ratio_hi(u32 j) const { return tab[5 * W + j]; }
    __device__ u64 matrix(u32 j, u32 i) const { return tab[(6 + j) * W + i]; }
};
constexpr size_t kConvTableWords = (size_t)(6 + kMaxWideLimbs) * kMaxWideLimbs;

namespace {

constexpr int kThreads = 256;

u32 grid_for(u64 items) {
    u64 g = (items + kThreads - 1) / kThreads;
    if (g == 0) g = 1;
    if (g > 0x7fffffffull) g = 0x7fffffffull;
    return (u32)g;
}

// reduce_dot_product (compact/slice.rs:380-405): one 128-bit accumulator (overflow discarded like the reference's
// carrying_add), reduce, reduce_add(., 0).  The reference folds the accumulator every DOT_PRODUCT_INNER_CHUNK = 16 terms;
// a sum of up to 32 products below 2^124 cannot overflow 128 bits, and the value reduced is the same integer.
template <class CT>
__device__ __forceinline__ u64 dot_mod(const CT &C, u32 j, const u64 (&t)[CT::kScaled]) {
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < CT::kScaled; ++i) {
        if ((u32)i < C.lin) {
            const u64 m = C.matrix(j, i);
            const u64 pl = t[i] * m, ph = mulhi64(t[i], m);
            lo += pl;
            hi += ph + (lo < pl);
        }
    }
    return add_mod(barrett_reduce128(lo, hi, C.p_out(j), C.ratio_lo(j), C.ratio_hi(j)), 0, C.p_out(j));
}

template <class CT, class WT>
__device__ __forceinline__ void load_scaled(const CT &C, const WT *__restrict__ in, u64 n, u64 c, u64 (&t)[CT::kScaled]) {
    // converter.rs:160-176: x mod q_i when the factor is one, else the Shoup product — both are
    // (factor * x) mod q_i, canonical
#pragma unroll
    for (int i = 0; i < CT::kScaled; ++i)
        t[i] = (u32)i < C.lin ? mul_shoup(in[(u64)i * n + c], C.inv_in(i), C.inv_in_p(i), C.q_in(i)) : 0;
}

// WT: the word type of the caller's arrays (u64, or u32 for BaseConverter<u32>: the same arithmetic on 32-bit words in
// memory; every result is canonical, so it is the integer the reference's u32 arithmetic produces)
template <class CT, class WT>
__global__ __launch_bounds__(kThreads) void fast_convert_kernel(CT C, const WT *__restrict__ in,
                                                                WT *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[CT::kScaled];
    load_scaled(C, in, n, c, t);
    for (u32 j = 0; j < C.lout; ++j) out[(u64)j * n + c] = (WT)dot_mod(C, j, t);
}

// fast_convert_array_to_pair_iter (converter.rs:233-272): two output moduli, one (mod p_0, mod p_1) pair per
// coefficient, written interleaved — one 16-byte (u32: 8-byte) store per thread
template <class CT, class WT>
__global__ __launch_bounds__(kThreads) void fast_convert_pair_kernel(CT C, const WT *__restrict__ in,
                                                                     WT *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[CT::kScaled];
    load_scaled(C, in, n, c, t);
    if constexpr (sizeof(WT) == 8) *reinterpret_cast<ulonglong2 *>(out + 2 * c) = ulonglong2{dot_mod(C, 0, t), dot_mod(C, 1, t)};
    else *reinterpret_cast<uint2 *>(out + 2 * c) = uint2{(u32)dot_mod(C, 0, t), (u32)dot_mod(C, 1, t)};
}

template <class CT, class WT>
__global__ __launch_bounds__(kThreads) void exact_convert_kernel(CT C, const WT *__restrict__ in,
                                                                 WT *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[CT::kScaled];
    load_scaled(C, in, n, c, t);
    double sum = 0.0;  // left to right, as converter.rs:291-345 sums
#pragma unroll
    for (int i = 0; i < CT::kScaled; ++i)
        if ((u32)i < C.lin) sum = __dadd_rn(sum, __ddiv_rn((double)t[i], (double)C.q_in(i)));
    const double r = __dadd_rn(sum, 0.5);
    u64 v;  // Rust `as u64` / `as u32`: truncation toward zero, saturating, NaN -> 0
    if (!(r > 0.0)) v = 0;
    else if (sizeof(WT) == 4 && r >= 4294967296.0) v = 0xffffffffull;
    else if (r >= 18446744073709551616.0) v = ~0ull;
    else v = (u64)r;
    const u64 dot = dot_mod(C, 0, t);
    const u64 vq = mul_mod_barrett(v, C.q_mod_p0, C.p_out(0), C.ratio_lo(0), C.ratio_hi(0));
    out[c] = (WT)sub_mod(dot, vq, C.p_out(0));
}

void barrett_ratio(u64 q, u64 &lo, u64 &hi) {
    unsigned __int128 rem = 1;
    unsigned __int128 c1 = rem << 64;
    hi = (u64)(c1 / q);
    rem = c1 % q;
    lo = (u64)((rem << 64) / q);
}

u64 big_mod(const u64 *limbs, u32 len, u64 q) {
    unsigned __int128 r = 0;
    for (u32 i = len; i-- > 0;) r = ((r << 64) | limbs[i]) % q;
    return (u64)r;
}

}  // namespace
}  // namespace pfhe

using namespace pfhe;

namespace pfhe {
struct ConvCore {
    int device = 0;
    u32 lin = 0, lout = 0;
    ConvDev dev{};                     // valid when both bases have at most kMaxLimbs moduli
    std::shared_ptr<DeviceBlob> blob;  // else: the device table of ConvWide
    u64 q_mod_p0 = 0;
    std::vector<u64> matrix;           // host copy, [lout][lin]
    bool wide() const { return blob != nullptr; }
};
}  // namespace pfhe

struct pfhe_conv {
    ConvCore c;
};
struct pfhe_conv32 {
    ConvCore c;
};

namespace {

// table of the device-table form
int conv_upload(int device, const RnsHost *in, const RnsHost &out, const std::vector<u64> &matrix,
                std::shared_ptr<DeviceBlob> &blob) {
    constexpr size_t W = kMaxWideLimbs;
    std::vector<u64> t(kConvTableWords, 0);
    const size_t lin = in ? in->moduli.size() : 0, lout = out.moduli.size();
    for (size_t i = 0; i < lin; ++i) {
        t[i] = in->moduli[i];
        t[W + i] = in->inv_punct[i];
        t[2 * W + i] = in->inv_punct_p[i];
    }
    for (size_t j = 0; j < lout; ++j) {
        t[3 * W + j] = out.moduli[j];
        barrett_ratio(out.moduli[j], t[4 * W + j], t[5 * W + j]);
        for (size_t i = 0; i < lin; ++i) t[(6 + j) * W + i] = matrix[j * lin + i];
    }
    auto b = std::make_shared<DeviceBlob>();
    b->device = device;
    PFHE_HIP(counted_malloc(&b->ptr, t.size() * sizeof(u64)));
    PFHE_HIP(hipMemcpy(b->ptr, t.data(), t.size() * sizeof(u64), hipMemcpyHostToDevice));
    blob = std::move(b);
    return PFHE_OK;
}

// runs F<CT>::launch with the constants in the form the converter holds
template <template <class> class F, class... A>
int conv_dispatch(const ConvCore *c, A &&...a) {
    if (!c->wide()) return F<ConvDev>::launch(c->dev, a...);
    const u64 *tab = (const u64 *)c->blob->ptr;
    if (c->lin <= 8) return F<ConvWide<8>>::launch(ConvWide<8>{c->lin, c->lout, tab, c->q_mod_p0}, a...);
    if (c->lin <= 16) return F<ConvWide<16>>::launch(ConvWide<16>{c->lin, c->lout, tab, c->q_mod_p0}, a...);
    if (c->lin <= 24) return F<ConvWide<24>>::launch(ConvWide<24>{c->lin, c->lout, tab, c->q_mod_p0}, a...);
    return F<ConvWide<32>>::launch(ConvWide<32>{c->lin, c->lout, tab, c->q_mod_p0}, a...);
}
template <class WT>
struct ConvLaunch {
    template <class CT>
    struct Fast {
        static int launch(const CT &C, const WT *in, WT *out, u64 n, hipStream_t s) {
            hipLaunchKernelGGL((fast_convert_kernel<CT, WT>), dim3(grid_for(n)), dim3(kThreads), 0, s, C, in, out, n);
            return PFHE_OK;
        }
    };
    template <class CT>
    struct Pair {
        static int launch(const CT &C, const WT *in, WT *out, u64 n, hipStream_t s) {
            hipLaunchKernelGGL((fast_convert_pair_kernel<CT, WT>), dim3(grid_for(n)), dim3(kThreads), 0, s, C, in, out, n);
            return PFHE_OK;
        }
    };
    template <class CT>
    struct Exact {
        static int launch(const CT &C, const WT *in, WT *out, u64 n, hipStream_t s) {
            hipLaunchKernelGGL((exact_convert_kernel<CT, WT>), dim3(grid_for(n)), dim3(kThreads), 0, s, C, in, out, n);
            return PFHE_OK;
        }
    };
};

// BaseConverter::new — converter.rs:43-69
int conv_create(const RnsHost &in, const RnsHost &ob, ConvCore *c) {
    if (in.device != ob.device) {
        set_last_error("input and output bases live on different devices");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    const u32 lin = (u32)in.moduli.size(), lout = (u32)ob.moduli.size(), len = in.par.dev.value_len;
    c->device = in.device;
    c->lin = lin;
    c->lout = lout;
    c->matrix.assign((size_t)lin * lout, 0);
    for (u32 j = 0; j < lout; ++j)
        for (u32 i = 0; i < lin; ++i) c->matrix[(size_t)j * lin + i] = big_mod(&in.punct[(size_t)i * len], len, ob.moduli[j]);
    c->q_mod_p0 = big_mod(in.Q.data(), len, ob.moduli[0]);
    if (lin <= (u32)kMaxLimbs && lout <= (u32)kMaxLimbs) {
        ConvDev &d = c->dev;
        d.lin = lin;
        d.lout = lout;
        for (u32 i = 0; i < lin; ++i) {
            d.q[i] = in.moduli[i];
            d.inv[i] = in.inv_punct[i];
            d.inv_p[i] = in.inv_punct_p[i];
        }
        for (u32 j = 0; j < lout; ++j) {
            d.p[j] = ob.moduli[j];
            barrett_ratio(ob.moduli[j], d.mu_lo[j], d.mu_hi[j]);
            for (u32 i = 0; i < lin; ++i) d.m[j][i] = c->matrix[(size_t)j * lin + i];
        }
        d.q_mod_p0 = c->q_mod_p0;
    } else {
        DeviceGuard g(c->device);
        if (!g.ok) return PFHE_ERR_NO_DEVICE;
        PFHE_TRY(conv_upload(c->device, &in, ob, c->matrix, c->blob));
    }
    return PFHE_OK;
}

template <class W>
int conv_matrix(const ConvCore *c, W *out, size_t len) {
    if (!c || !out) return PFHE_ERR_BAD_ARGUMENT;
    if (len != c->matrix.size()) return PFHE_ERR_BAD_LENGTH;
    for (size_t i = 0; i < len; ++i) out[i] = (W)c->matrix[i];
    return PFHE_OK;
}

int conv_check(const ConvCore *c, const void *in, size_t len_in, const void *out, size_t len_out, size_t poly_length,
               bool exact) {
    if (!c || ((!in || !out) && poly_length)) return PFHE_ERR_BAD_ARGUMENT;
    if (exact && c->lout != 1) {  // converter.rs:284-288 asserts
        set_last_error("output base in exact_convert_array must hold exactly one modulus");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    if (len_in != poly_length * c->lin || len_out != poly_length * c->lout) return PFHE_ERR_BAD_LENGTH;
    return PFHE_OK;
}

template <class W>
int conv_array_dev(const ConvCore *c, const W *in_dev, size_t len_in, W *out_dev, size_t len_out, size_t poly_length,
                   bool exact, void *stream) {
    PFHE_TRY(conv_check(c, in_dev, len_in, out_dev, len_out, poly_length, exact));
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    if (exact) {
        PFHE_TRY((conv_dispatch<ConvLaunch<W>::template Exact>(c, in_dev, out_dev, (u64)poly_length, (hipStream_t)stream)));
    } else {
        PFHE_TRY((conv_dispatch<ConvLaunch<W>::template Fast>(c, in_dev, out_dev, (u64)poly_length, (hipStream_t)stream)));
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class W>
int conv_pairs_dev(const ConvCore *c, const W *in_dev, size_t len_in, W *pairs_out_dev, size_t len_out, size_t poly_length,
                   void *stream) {
    if (!c || ((!in_dev || !pairs_out_dev) && poly_length)) return PFHE_ERR_BAD_ARGUMENT;
    if (c->lout != 2) {  // converter.rs:239-243 asserts
        set_last_error("output base in fast_convert_array_to_pair must contain exactly two moduli");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    if (len_in != poly_length * c->lin || len_out != 2 * poly_length) return PFHE_ERR_BAD_LENGTH;
    PFHE_REQUIRE_ALIGNED(pairs_out_dev);
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    PFHE_TRY((conv_dispatch<ConvLaunch<W>::template Pair>(c, in_dev, pairs_out_dev, (u64)poly_length, (hipStream_t)stream)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// host-pointer forms: stage, run, copy back
template <class W>
int conv_host(const ConvCore *c, const W *in, size_t len_in, W *out, size_t len_out, size_t poly_length, bool exact) {
    PFHE_TRY(conv_check(c, in, len_in, out, len_out, poly_length, exact));
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(c->device);  // pooled staging context: no allocation in steady state
    if (!st.ok()) return PFHE_ERR_HIP;
    void *din = nullptr, *dout = nullptr;
    PFHE_TRY(st.upload(in, len_in * sizeof(W), &din));
    PFHE_TRY(st.alloc(len_out * sizeof(W), &dout));
    PFHE_TRY(conv_array_dev<W>(c, (const W *)din, len_in, (W *)dout, len_out, poly_length, exact, st.stream()));
    PFHE_TRY(st.download(out, dout, len_out * sizeof(W)));
    return st.finish();
}

}  // namespace

// the C entry points of one word type: CV = pfhe_conv / pfhe_conv32, NS = pfhe_rns / pfhe_rns32
#define PFHE_CONV_FAMILY(CV, NS, W)                                                                                       \
    int CV##_create(const NS *input_base, const NS *output_base, CV **out) {                                              \
        PFHE_GUARD_BEGIN                                                                                                  \
        if (!input_base || !output_base || !out) return PFHE_ERR_BAD_ARGUMENT;                                            \
        *out = nullptr;                                                                                                   \
        auto c = std::make_unique<CV>();                                                                                  \
        PFHE_TRY(conv_create(input_base->h, output_base->h, &c->c));                                                      \
        *out = c.release();                                                                                               \
        return PFHE_OK;                                                                                                   \
        PFHE_GUARD_END                                                                                                    \
    }                                                                                                                     \
    void CV##_destroy(CV *c) { delete c; }                                                                                \
    size_t CV##_input_moduli_count(const CV *c) { return c ? c->c.lin : 0; }                                              \
    size_t CV##_output_moduli_count(const CV *c) { return c ? c->c.lout : 0; }                                            \
    int CV##_base_change_matrix(const CV *c, W *out, size_t len) { return conv_matrix<W>(c ? &c->c : nullptr, out, len); } \
    int CV##_fast_convert_array_dev(const CV *c, const W *crt_poly_in_dev, size_t len_in, W *crt_poly_out_dev,            \
                                    size_t len_out, size_t poly_length, void *stream) {                                   \
        PFHE_GUARD_BEGIN                                                                                                  \
        return conv_array_dev<W>(c ? &c->c : nullptr, crt_poly_in_dev, len_in, crt_poly_out_dev, len_out, poly_length,    \
                                 false, stream);                                                                          \
        PFHE_GUARD_END                                                                                                    \
    }                                                                                                                     \
    int CV##_fast_convert_array_to_pairs_dev(const CV *c, const W *crt_poly_in_dev, size_t len_in, W *pairs_out_dev,      \
                                             size_t len_out, size_t poly_length, void *stream) {                          \
        PFHE_GUARD_BEGIN                                                                                                  \
        return conv_pairs_dev<W>(c ? &c->c : nullptr, crt_poly_in_dev, len_in, pairs_out_dev, len_out, poly_length,       \
                                 stream);                                                                                 \
        PFHE_GUARD_END                                                                                                    \
    }                                                                                                                     \
    int CV##_exact_convert_array_dev(const CV *c, const W *crt_poly_in_dev, size_t len_in, W *crt_poly_out_dev,           \
                                     size_t len_out, size_t poly_length, void *stream) {                                  \
        PFHE_GUARD_BEGIN                                                                                                  \
        return conv_array_dev<W>(c ? &c->c : nullptr, crt_poly_in_dev, len_in, crt_poly_out_dev, len_out, poly_length,    \
                                 true, stream);                                                                           \
        PFHE_GUARD_END                                                                                                    \
    }                                                                                                                     \
    int CV##_fast_convert_array(const CV *c, const W *crt_poly_in, size_t len_in, W *crt_poly_out, size_t len_out,        \
                                size_t poly_length) {                                                                     \
        PFHE_GUARD_BEGIN                                                                                                  \
        return conv_host<W>(c ? &c->c : nullptr, crt_poly_in, len_in, crt_poly_out, len_out, poly_length, false);         \
        PFHE_GUARD_END                                                                                                    \
    }                                                                                                                     \
    int CV##_exact_convert_array(const CV *c, const W *crt_poly_in, size_t len_in, W *crt_poly_out, size_t len_out,       \
                                 size_t poly_length) {                                                                    \
        PFHE_GUARD_BEGIN                                                                                                  \
        return conv_host<W>(c ? &c->c : nullptr, crt_poly_in, len_in, crt_poly_out, len_out, poly_length, true);          \
        PFHE_GUARD_END                                                                                                    \
    }

extern "C" {
PFHE_CONV_FAMILY(pfhe_conv, pfhe_rns, uint64_t)
PFHE_CONV_FAMILY(pfhe_conv32, pfhe_rns32, uint32_t)
}  // extern "C"
