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

struct ConvDev {
    u32 lin, lout;
    u64 q[kMaxLimbs], inv[kMaxLimbs], inv_p[kMaxLimbs];       // input base
    u64 p[kMaxLimbs], mu_lo[kMaxLimbs], mu_hi[kMaxLimbs];     // output base + floor(2^128/p)
    u64 m[kMaxLimbs][kMaxLimbs];                              // m[j][i] = (Q/q_i) mod p_j
    u64 q_mod_p0;                                             // Q mod p_0
};

namespace {

constexpr int kThreads = 256;

u32 grid_for(u64 items) {
    u64 g = (items + kThreads - 1) / kThreads;
    if (g == 0) g = 1;
    if (g > 0x7fffffffull) g = 0x7fffffffull;
    return (u32)g;
}

// reduce_dot_product for fewer than DOT_PRODUCT_INNER_CHUNK = 16 terms (compact/slice.rs:380-405):
// one 128-bit accumulator (overflow discarded like the reference's carrying_add), reduce, reduce_add(., 0)
__device__ __forceinline__ u64 dot_mod(const ConvDev &C, u32 j, const u64 *t) {
    u64 lo = 0, hi = 0;
    for (u32 i = 0; i < C.lin; ++i) {
        const u64 pl = t[i] * C.m[j][i], ph = mulhi64(t[i], C.m[j][i]);
        lo += pl;
        hi += ph + (lo < pl);
    }
    return add_mod(barrett_reduce128(lo, hi, C.p[j], C.mu_lo[j], C.mu_hi[j]), 0, C.p[j]);
}

__device__ __forceinline__ void load_scaled(const ConvDev &C, const u64 *__restrict__ in, u64 n, u64 c, u64 *t) {
    // converter.rs:160-176: x mod q_i when the factor is one, else the Shoup product — both are
    // (factor * x) mod q_i, canonical
    for (u32 i = 0; i < C.lin; ++i) t[i] = mul_shoup(in[(u64)i * n + c], C.inv[i], C.inv_p[i], C.q[i]);
}

__global__ __launch_bounds__(kThreads) void fast_convert_kernel(ConvDev C, const u64 *__restrict__ in,
                                                                u64 *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[kMaxLimbs];
    load_scaled(C, in, n, c, t);
    for (u32 j = 0; j < C.lout; ++j) out[(u64)j * n + c] = dot_mod(C, j, t);
}

// fast_convert_array_to_pair_iter (converter.rs:233-272): two output moduli, one (mod p_0, mod p_1) pair per
// coefficient, written interleaved — one 16-byte store per thread
__global__ __launch_bounds__(kThreads) void fast_convert_pair_kernel(ConvDev C, const u64 *__restrict__ in,
                                                                     u64 *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[kMaxLimbs];
    load_scaled(C, in, n, c, t);
    *reinterpret_cast<ulonglong2 *>(out + 2 * c) = ulonglong2{dot_mod(C, 0, t), dot_mod(C, 1, t)};
}

__global__ __launch_bounds__(kThreads) void exact_convert_kernel(ConvDev C, const u64 *__restrict__ in,
                                                                 u64 *__restrict__ out, u64 n) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    u64 t[kMaxLimbs];
    load_scaled(C, in, n, c, t);
    double sum = 0.0;
    for (u32 i = 0; i < C.lin; ++i) sum = __dadd_rn(sum, __ddiv_rn((double)t[i], (double)C.q[i]));
    const double r = __dadd_rn(sum, 0.5);
    u64 v;  // Rust `as u64`: truncation toward zero, saturating, NaN -> 0
    if (!(r > 0.0)) v = 0;
    else if (r >= 18446744073709551616.0) v = ~0ull;
    else v = (u64)r;
    const u64 dot = dot_mod(C, 0, t);
    const u64 vq = mul_mod_barrett(v, C.q_mod_p0, C.p[0], C.mu_lo[0], C.mu_hi[0]);
    out[c] = sub_mod(dot, vq, C.p[0]);
}

// value mod q_i by Horner over the limbs, most significant first; every step reduces hi:lo < q*2^64
__global__ __launch_bounds__(kThreads) void decompose_big_kernel(ConvDev C, u32 value_len,
                                                                 const u64 *__restrict__ values,
                                                                 u64 *__restrict__ multi, u64 count) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    u64 v[kMaxLimbs];
    for (u32 k = 0; k < value_len; ++k) v[k] = values[c * value_len + k];
    for (u32 j = 0; j < C.lout; ++j) {
        u64 r = 0;
        for (u32 k = value_len; k-- > 0;) r = barrett_reduce128(v[k], r, C.p[j], C.mu_lo[j], C.mu_hi[j]);
        multi[(u64)j * count + c] = r;
    }
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

void fill_output_side(ConvDev &c, const RnsDev &out) {
    c.lout = out.L;
    for (u32 j = 0; j < out.L; ++j) {
        c.p[j] = out.q[j];
        barrett_ratio(out.q[j], c.mu_lo[j], c.mu_hi[j]);
    }
}

}  // namespace
}  // namespace pfhe

using namespace pfhe;

struct pfhe_conv {
    int device = 0;
    ConvDev dev{};
};

extern "C" {

int pfhe_conv_create(const pfhe_rns *input_base, const pfhe_rns *output_base, pfhe_conv **out) {
    PFHE_GUARD_BEGIN
    if (!input_base || !output_base || !out) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (input_base->h.device != output_base->h.device) {
        set_last_error("input and output bases live on different devices");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    auto c = std::make_unique<pfhe_conv>();
    const RnsDev &in = input_base->h.dev;
    c->device = input_base->h.device;
    c->dev.lin = in.L;
    for (u32 i = 0; i < in.L; ++i) {
        c->dev.q[i] = in.q[i];
        c->dev.inv[i] = in.inv_punct[i];
        c->dev.inv_p[i] = in.inv_punct_p[i];
    }
    fill_output_side(c->dev, output_base->h.dev);
    for (u32 j = 0; j < c->dev.lout; ++j)
        for (u32 i = 0; i < in.L; ++i) c->dev.m[j][i] = big_mod(in.punct[i], in.value_len, c->dev.p[j]);
    c->dev.q_mod_p0 = big_mod(in.Q, in.value_len, c->dev.p[0]);
    *out = c.release();
    return PFHE_OK;
    PFHE_GUARD_END
}

void pfhe_conv_destroy(pfhe_conv *c) { delete c; }
size_t pfhe_conv_input_moduli_count(const pfhe_conv *c) { return c ? c->dev.lin : 0; }
size_t pfhe_conv_output_moduli_count(const pfhe_conv *c) { return c ? c->dev.lout : 0; }

int pfhe_conv_base_change_matrix(const pfhe_conv *c, uint64_t *out, size_t len) {
    if (!c || !out) return PFHE_ERR_BAD_ARGUMENT;
    if (len != (size_t)c->dev.lin * c->dev.lout) return PFHE_ERR_BAD_LENGTH;
    for (u32 j = 0; j < c->dev.lout; ++j)
        for (u32 i = 0; i < c->dev.lin; ++i) out[j * c->dev.lin + i] = c->dev.m[j][i];
    return PFHE_OK;
}

static int conv_check(const pfhe_conv *c, const void *in, size_t len_in, const void *out, size_t len_out,
                      size_t poly_length, bool exact) {
    if (!c || ((!in || !out) && poly_length)) return PFHE_ERR_BAD_ARGUMENT;
    if (exact && c->dev.lout != 1) {  // converter.rs:284-288 asserts
        set_last_error("output base in exact_convert_array must hold exactly one modulus");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    if (len_in != poly_length * c->dev.lin || len_out != poly_length * c->dev.lout) return PFHE_ERR_BAD_LENGTH;
    return PFHE_OK;
}

int pfhe_conv_fast_convert_array_dev(const pfhe_conv *c, const uint64_t *crt_poly_in_dev, size_t len_in,
                                     uint64_t *crt_poly_out_dev, size_t len_out, size_t poly_length, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(conv_check(c, crt_poly_in_dev, len_in, crt_poly_out_dev, len_out, poly_length, false));
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    hipLaunchKernelGGL(fast_convert_kernel, dim3(grid_for(poly_length)), dim3(kThreads), 0, (hipStream_t)stream, c->dev,
                       (const u64 *)crt_poly_in_dev, (u64 *)crt_poly_out_dev, (u64)poly_length);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_conv_fast_convert_array_to_pairs_dev(const pfhe_conv *c, const uint64_t *crt_poly_in_dev, size_t len_in,
                                              uint64_t *pairs_out_dev, size_t len_out, size_t poly_length, void *stream) {
    PFHE_GUARD_BEGIN
    if (!c || ((!crt_poly_in_dev || !pairs_out_dev) && poly_length)) return PFHE_ERR_BAD_ARGUMENT;
    if (c->dev.lout != 2) {  // converter.rs:239-243 asserts
        set_last_error("output base in fast_convert_array_to_pair must contain exactly two moduli");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    if (len_in != poly_length * c->dev.lin || len_out != 2 * poly_length) return PFHE_ERR_BAD_LENGTH;
    PFHE_REQUIRE_ALIGNED(pairs_out_dev);
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    hipLaunchKernelGGL(fast_convert_pair_kernel, dim3(grid_for(poly_length)), dim3(kThreads), 0, (hipStream_t)stream,
                       c->dev, (const u64 *)crt_poly_in_dev, (u64 *)pairs_out_dev, (u64)poly_length);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_conv_exact_convert_array_dev(const pfhe_conv *c, const uint64_t *crt_poly_in_dev, size_t len_in,
                                      uint64_t *crt_poly_out_dev, size_t len_out, size_t poly_length, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(conv_check(c, crt_poly_in_dev, len_in, crt_poly_out_dev, len_out, poly_length, true));
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    hipLaunchKernelGGL(exact_convert_kernel, dim3(grid_for(poly_length)), dim3(kThreads), 0, (hipStream_t)stream, c->dev,
                       (const u64 *)crt_poly_in_dev, (u64 *)crt_poly_out_dev, (u64)poly_length);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

// host-pointer forms: stage, run, copy back
static int conv_host(const pfhe_conv *c, const uint64_t *in, size_t len_in, uint64_t *out, size_t len_out,
                     size_t poly_length, bool exact) {
    PFHE_TRY(conv_check(c, in, len_in, out, len_out, poly_length, exact));
    if (poly_length == 0) return PFHE_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(c->device);  // pooled staging context: no allocation in steady state
    if (!st.ok()) return PFHE_ERR_HIP;
    void *din = nullptr, *dout = nullptr;
    PFHE_TRY(st.upload(in, len_in * 8, &din));
    PFHE_TRY(st.alloc(len_out * 8, &dout));
    PFHE_TRY(exact ? pfhe_conv_exact_convert_array_dev(c, (const uint64_t *)din, len_in, (uint64_t *)dout, len_out,
                                                       poly_length, st.stream())
                   : pfhe_conv_fast_convert_array_dev(c, (const uint64_t *)din, len_in, (uint64_t *)dout, len_out,
                                                      poly_length, st.stream()));
    PFHE_TRY(st.download(out, dout, len_out * 8));
    return st.finish();
}

int pfhe_conv_fast_convert_array(const pfhe_conv *c, const uint64_t *crt_poly_in, size_t len_in, uint64_t *crt_poly_out,
                                 size_t len_out, size_t poly_length) {
    PFHE_GUARD_BEGIN
    return conv_host(c, crt_poly_in, len_in, crt_poly_out, len_out, poly_length, false);
    PFHE_GUARD_END
}

int pfhe_conv_exact_convert_array(const pfhe_conv *c, const uint64_t *crt_poly_in, size_t len_in,
                                  uint64_t *crt_poly_out, size_t len_out, size_t poly_length) {
    PFHE_GUARD_BEGIN
    return conv_host(c, crt_poly_in, len_in, crt_poly_out, len_out, poly_length, true);
    PFHE_GUARD_END
}

int pfhe_rns_decompose_big_uint_values_to_dev(const pfhe_rns *r, const uint64_t *big_uint_values_dev, size_t len_in,
                                              uint64_t *multi_residues_dev, size_t len_out, size_t value_count,
                                              void *stream) {
    PFHE_GUARD_BEGIN
    if (!r || ((!big_uint_values_dev || !multi_residues_dev) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    const RnsDev &R = r->h.dev;
    if (len_in != value_count * R.value_len || len_out != value_count * R.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(r->h.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    ConvDev c{};
    fill_output_side(c, R);
    hipLaunchKernelGGL(decompose_big_kernel, dim3(grid_for(value_count)), dim3(kThreads), 0, (hipStream_t)stream, c,
                       R.value_len, (const u64 *)big_uint_values_dev, (u64 *)multi_residues_dev, (u64)value_count);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_rns_decompose_big_uint_values_to(const pfhe_rns *r, const uint64_t *big_uint_values, size_t len_in,
                                          uint64_t *multi_residues, size_t len_out, size_t value_count) {
    PFHE_GUARD_BEGIN
    if (!r || ((!big_uint_values || !multi_residues) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_in != value_count * r->h.dev.value_len || len_out != value_count * r->h.dev.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(r->h.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(r->h.device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *din = nullptr, *dout = nullptr;
    PFHE_TRY(st.upload(big_uint_values, len_in * 8, &din));
    PFHE_TRY(st.alloc(len_out * 8, &dout));
    PFHE_TRY(pfhe_rns_decompose_big_uint_values_to_dev(r, (const uint64_t *)din, len_in, (uint64_t *)dout, len_out,
                                                       value_count, st.stream()));
    PFHE_TRY(st.download(multi_residues, dout, len_out * 8));
    return st.finish();
    PFHE_GUARD_END
}

}  // extern "C"
