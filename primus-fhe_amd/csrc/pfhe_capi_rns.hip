// pfhe_capi_rns.hip — extern "C" boundary for RNSBase, BigUintApproxSignedBasis and the RNS gadget
// external product (include/pfhe.h, second half).
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>

#include <cstring>
#include <type_traits>

#include "pfhe_capi_internal.hpp"
#include "pfhe_ntt_device.hpp"
#include "pfhe_rns.hpp"
#include "pfhe_staging.hpp"

using namespace pfhe;

struct pfhe_extprod_plan {
    const TableSet *table = nullptr;  // borrowed from the pfhe_dcrt (must outlive the plan)
    // Exclusivity.  The plan owns the product's scratch (digit buffers), like the reference's `&mut DcrtGlevContext`
    // (primus_lattice/src/context/glev.rs:4-10), which the borrow checker lets ONE caller hold at a time.  Here the holder
    // is a thread: every entry point that touches the scratch takes the plan for the duration of the call (PlanLease), and
    // a second thread that arrives meanwhile is refused with PFHE_ERR_BUSY ("plan in use") instead of racing on the
    // digit buffer.  owner = a per-thread token (0: free); depth counts nested entries of the owning thread (the host-pointer
    // and profiling entry points call the device ones).
    std::atomic<std::uintptr_t> owner{0};
    int depth = 0;
    // cross-stream ordering of successive calls (run_product): the event recorded behind the last call's kernels
    hipEvent_t last_done = nullptr;
    bool last_valid = false;
    RnsParams rns;
    BasisParams basis_par;
    BasisCore basis{};  // the scalar constants of basis_par
    u32 k = 1;
    size_t chunk = 1;
    // digit buffer of chunk * (k+1) * ell * L * N words.  Chunks run one after the other on the caller's stream (every
    // kernel of the product is bound by the same integer ALU: a two-stream software pipeline over two buffers measured
    // 20.9 ms per 1024 products at N = 2^16 against 20.4 ms in order, and is gone).
    // PFHE_DISABLE_FUSED_EXTPROD, read at plan creation: the unfused transform + multiply-accumulate kernels (the form
    // k > 1 takes) for every shape — the parity tests compare the two
    bool use_fused = true;
    // workgroups a call must offer before the fused block + multiply-accumulate kernel is taken (N = 2^16, 3 limbs,
    // coefficient form: 1 / 2 / 4 / 5 ciphertexts take 97 / 121 / 165 / 190 us unfused and 141 / 144 / 158 / 164 us fused)
    static constexpr u64 fused_min_wgs = 160;
    // measurement aid (pfhe_extprod_profile_dev): when non-null, run_product records an event before the
    // decomposition, between the decomposition and the transform / multiply-accumulate, and after it, per chunk
    std::vector<hipEvent_t> *prof = nullptr;
    u64 *digits = nullptr;
    size_t digits_words = 0;
    void *sdigits = nullptr;  // compact signed digits of one chunk (chunk * (k+1) * ell * N words of sdigit_bytes: int32 when log_basis <= 31, else int64), or null
    size_t sdigit_bytes = 0;
    ~pfhe_extprod_plan() {
        if (!table) return;
        DeviceGuard g(table->device);
        if (digits) (void)counted_free(digits);
        if (sdigits) (void)counted_free(sdigits);
        if (last_done) (void)hipEventDestroy(last_done);
    }
};

// The <u32> instantiation of the same product: CrtGlwe<u32>::mul_dcrt_ggsw_to over a U32DcrtTable (glwe/crt.rs:200-227,
// dcrt/prime32.rs:11).  N = 2^16, k = 1 (the bench shape) takes the 64-bit plan's kernels on B32Arith: balanced int32
// digits, lift + strided pass, block pass + multiply-accumulate (+ inverse block pass) with the transformed digits on chip.
// Every other shape: steps (1)-(4) fused (gadget_decompose_kernel on u32 words), the table's forward transform over the
// lifted digit polynomials, one multiply-accumulate kernel.  Chunk after chunk on the caller's stream.
struct pfhe_extprod32_plan {
    const TableSet *table = nullptr;  // borrowed from the pfhe_dcrt32 (must outlive the plan)
    std::atomic<std::uintptr_t> owner{0};  // one holder at a time, as pfhe_extprod_plan
    int depth = 0;
    hipEvent_t last_done = nullptr;
    bool last_valid = false;
    RnsParams rns;
    BasisParams basis_par;
    BasisCore basis{};
    u32 k = 1;
    size_t chunk = 1;
    u32 *digits = nullptr;  // chunk * (k+1) * ell * L * N words
    size_t digits_words = 0;
    int *sdigits = nullptr;  // N = 2^16, k = 1 (the fused kernels): chunk * (k+1) * ell * N balanced digits
    bool use_fused = true;   // PFHE_DISABLE_FUSED_EXTPROD, read at plan creation
    ~pfhe_extprod32_plan() {
        if (!table) return;
        DeviceGuard g(table->device);
        if (digits) (void)counted_free(digits);
        if (sdigits) (void)counted_free(sdigits);
        if (last_done) (void)hipEventDestroy(last_done);
    }
};

namespace {

int plan_check(const pfhe_extprod_plan *p) {
    if (!p || !p->table) return PFHE_ERR_BAD_ARGUMENT;
    return PFHE_OK;
}

// the calling thread's hold on a plan's scratch for one entry point (see pfhe_extprod_plan::owner)
inline std::uintptr_t plan_thread_token() {
    static thread_local char token;
    return reinterpret_cast<std::uintptr_t>(&token);
}
// take (or re-enter) the plan for the calling thread; false: another thread holds it
template <class Plan>
bool plan_acquire(Plan *p) {
    const std::uintptr_t me = plan_thread_token();
    std::uintptr_t free_ = 0;
    if (p->owner.load(std::memory_order_relaxed) == me) {
        ++p->depth;  // nested entry of the thread that holds the plan
        return true;
    }
    if (p->owner.compare_exchange_strong(free_, me, std::memory_order_acquire)) {
        p->depth = 1;
        return true;
    }
    return false;
}
template <class Plan>
void plan_release(Plan *p) {
    if (--p->depth == 0) p->owner.store(0, std::memory_order_release);
}
template <class Plan>
class PlanLease {
  public:
    explicit PlanLease(Plan *p) : p_(p), held_(plan_acquire(p)) {}
    ~PlanLease() {
        if (held_) plan_release(p_);
    }
    PlanLease(const PlanLease &) = delete;
    PlanLease &operator=(const PlanLease &) = delete;
    bool held() const { return held_; }

  private:
    Plan *p_;
    bool held_ = false;
};
#define PFHE_PLAN_LEASE(plan)                                                                                       \
    PlanLease<typename std::remove_pointer<decltype(plan)>::type> lease_(plan);                                                                                          \
    if (!lease_.held()) {                                                                                            \
        set_last_error("external-product plan in use by another thread (one plan per thread, like &mut DcrtGlevContext)"); \
        return PFHE_ERR_BUSY;                                                                                        \
    }

// one row of the product: acc[e] += glev[e or shared] (x) crt_poly[e]   (glwe/dcrt.rs:178-255)
// rows == k+1 without `accumulate` gives CrtGlwe::mul_dcrt_ggsw_to (glwe/crt.rs:200-227).
// Chunks of ciphertexts run one after the other on the caller's stream.
// `into_coeff`: the caller wants coefficient-form output; *coeff_passes reports how many passes of the inverse
// transform this function already ran on the result: -1 = all of them (small-ring kernel), 1 = the block pass
// (fused into the multiply-accumulate kernel; the caller runs the remaining strided pass), 0 = none.
// `big_input`: the input polynomials are BigUintPolynomials (value_len limbs per coefficient) instead of CRT ones.
int run_product_impl(pfhe_extprod_plan *p, const u64 *crt_polys, u32 rows, const u64 *keys, bool keys_shared, u64 *result,
                     u64 batch, bool accumulate, hipStream_t s, bool into_coeff, int *coeff_passes, bool big_input) {
    if (coeff_passes) *coeff_passes = 0;
    const TableSet &t = *p->table;
    const u64 W = (u64)t.L * t.n;
    RnsParams rns = p->rns;
    rns.dev.big_input = rns.wide_tab.big_input = big_input ? 1u : 0u;
    const u64 in_words = big_input ? (u64)rns.dev.value_len * t.n : W;  // words per input polynomial
    const u32 ell = p->basis.ell;
    const u64 key_words = (u64)rows * ell * (p->k + 1) * W;
    // Everything runs chunk after chunk on the caller's stream (also while the caller captures a HIP graph).
    // small rings: digit extraction + ONE kernel for everything else
    // (one workgroup per (ciphertext, limb) runs 12+ transforms back to back: it needs a batch that fills the chip)
    if (p->sdigits != nullptr && extprod_small_supported(t.log_n, p->k, p->rns.dev.value_len, p->basis.log_basis) &&
        batch * t.L >= 1024 && p->use_fused) {
        for (u64 done = 0; done < batch; done += p->chunk) {
            const u64 cur = std::min<u64>(p->chunk, batch - done);
            PFHE_TRY(gadget_signed_digits_dev(rns, p->basis_par, t.log_n, crt_polys + done * rows * in_words, (int *)p->sdigits, cur * rows, s));
            PFHE_TRY(extprod_small_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, p->k, rows, ell, (const int *)p->sdigits,
                                       keys + (keys_shared ? 0 : done * key_words), keys_shared,
                                       result + done * (p->k + 1) * W, cur, accumulate, into_coeff, s));
        }
        if (coeff_passes) *coeff_passes = into_coeff ? -1 : 0;
        return PFHE_OK;
    }
    const bool fused = gadget_fused_supported(t.log_n, p->k) && p->use_fused &&
                       ((std::min<u64>(batch, p->chunk) * t.L) << (t.log_n - 12)) >= p->fused_min_wgs;
    const int passes = ntt_num_passes(t.log_n, t.ntt_arith, t.tune);
    // One arithmetic for everything in here — the gadget kernels and the plain transform passes around them: the table's
    // transform arithmetic t.ntt_arith (pseudo-Mersenne, Montgomery form for generic primes below 2^61, or the
    // reference's Shoup form).
    // coefficient-form output: the inverse transform's block pass runs inside the fused kernel, on the accumulators
    const bool inv_tail = fused && into_coeff && !accumulate && coeff_passes != nullptr && passes == 2;
    if (inv_tail) *coeff_passes = 1;
    const bool fused_decompose = gadget_decompose_strided_supported(t.log_n, p->rns.dev.value_len) && p->sdigits != nullptr;
    u64 *dg = p->digits;
    for (u64 done = 0; done < batch; done += p->chunk) {
        const u64 cur = std::min<u64>(p->chunk, batch - done);
        const u64 npolys = cur * rows * ell * t.L;
        const auto stamp = [&]() -> int {
            if (p->prof == nullptr) return PFHE_OK;
            hipEvent_t ev = nullptr;
            PFHE_HIP(hipEventCreate(&ev));
            p->prof->push_back(ev);
            PFHE_HIP(hipEventRecord(ev, s));
            return PFHE_OK;
        };
        PFHE_TRY(stamp());
        // ---- steps (1)-(4) + strided passes into the digit buffer ----
        if (fused_decompose) {
            PFHE_TRY(gadget_decompose_strided_dev(rns, p->basis_par, t.primes_dev, t.log_n, t.ntt_arith,
                                                  crt_polys + done * rows * in_words, dg, cur * rows, s, p->sdigits));
        } else {
            PFHE_TRY(gadget_decompose_dev(rns, p->basis_par, t.log_n, crt_polys + done * rows * in_words, dg, cur * rows, s));
            for (int i = 0; i < passes - 1; ++i)
                PFHE_TRY(ntt_pass_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, dg, npolys, false, i, false, s, nullptr, 0, t.tune));
        }
        PFHE_TRY(stamp());
        // ---- block pass (last pass of the transform) + multiply-accumulate ----
        if (fused) {
            PFHE_TRY(gadget_block_mulacc_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, p->k, rows * ell, dg,
                                             keys + (keys_shared ? 0 : done * key_words), keys_shared,
                                             result + done * (p->k + 1) * W, cur, accumulate, s, inv_tail));
        } else {
            PFHE_TRY(ntt_pass_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, dg, npolys, false, passes - 1, false, s, nullptr, 0,
                                  t.tune));
            PFHE_TRY(gadget_mulacc_dev(t.primes_dev, t.L, t.log_n, p->k, rows, ell, dg,
                                       keys + (keys_shared ? 0 : done * key_words), keys_shared,
                                       result + done * (p->k + 1) * W, cur, accumulate, s));
        }
        PFHE_TRY(stamp());
    }
    return PFHE_OK;
}


// The plan's digit buffers are touched by run_product_impl only.  Successive calls on DIFFERENT streams are ordered here:
// every call records the plan's `last_done` event behind its last kernel, and a call on another stream first makes that
// stream wait for it — so "one plan, used from one stream after another" needs no event handling by the caller (calls by
// two THREADS at once are refused by the lease above; work captured into a HIP graph is outside this bookkeeping: a
// capturing stream neither waits nor records, and a graph that uses a plan must not be replayed beside other users of it).
int run_product(pfhe_extprod_plan *p, const u64 *crt_polys, u32 rows, const u64 *keys, bool keys_shared, u64 *result,
                u64 batch, bool accumulate, hipStream_t s, bool into_coeff = false, int *coeff_passes = nullptr,
                bool big_input = false) {
    const bool tracked = p->last_done != nullptr && !stream_is_capturing(s);
    // (always, also on the stream that recorded it: a handle comparison would miss a stream destroyed and re-created at
    // the same address; waiting on one's own stream's event costs nothing)
    if (tracked && p->last_valid) PFHE_HIP(hipStreamWaitEvent(s, p->last_done, 0));
    const int rc = run_product_impl(p, crt_polys, rows, keys, keys_shared, result, batch, accumulate, s, into_coeff, coeff_passes,
                                    big_input);
    if (tracked) {  // also after a failed call: whatever it queued still uses the buffers
        if (hipEventRecord(p->last_done, s) == hipSuccess) {
            p->last_valid = true;
        } else {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);  // no event: fall back to draining the stream
            p->last_valid = false;
        }
    }
    return rc;
}

}  // namespace

// ---- RNSBase<W> / BigUintApproxSignedBasis<W> behind both word widths of the C ABI (W = uint64_t: pfhe_rns / pfhe_basis;
//      W = uint32_t: pfhe_rns32 / pfhe_basis32).  One implementation, instantiated twice. ----
namespace {

template <class W>
using DevWord = typename std::conditional<sizeof(W) == 8, u64, u32>::type;
template <class W>
size_t words_per_value(const RnsHost &h) { return sizeof(W) == 8 ? h.par.dev.value_len : h.par.dev.value_words; }
template <class W>
size_t words_per_value(const BasisHost &h) { return sizeof(W) == 8 ? h.par.dev.value_len : h.par.dev.value_words; }

// word j (of the caller's width) of a big integer held as 64-bit limbs
template <class W>
W word_of(const std::vector<u64> &limbs, size_t base, size_t j) {
    if (sizeof(W) == 8) return (W)limbs[base + j];
    return (W)(limbs[base + j / 2] >> (32 * (j & 1)));
}

template <class W>
int rns_create_impl(const W *moduli, size_t count, int device, RnsHost &h) {
    if (!moduli && count) return PFHE_ERR_BAD_ARGUMENT;
    std::vector<u64> m(moduli, moduli + count);
    PFHE_TRY(build_rns(m.data(), count, h, 8 * sizeof(W)));
    PFHE_TRY(capi_check_device(device));
    h.device = device;
    if (h.par.wide()) {  // more than kMaxLimbs moduli: the constants live in a device table
        DeviceGuard g(device);
        if (!g.ok) return PFHE_ERR_NO_DEVICE;
        PFHE_TRY(upload_rns_wide(h));
    }
    return PFHE_OK;
}

template <class W>
int rns_moduli_product_impl(const RnsHost *h, W *out, size_t len) {
    if (!h || !out) return PFHE_ERR_BAD_ARGUMENT;
    if (len != words_per_value<W>(*h)) return PFHE_ERR_BAD_LENGTH;
    for (size_t j = 0; j < len; ++j) out[j] = word_of<W>(h->Q, 0, j);
    return PFHE_OK;
}

template <class W>
int rns_compose_dev_impl(const RnsHost *h, const W *in_dev, size_t len_in, W *out_dev, size_t len_out, size_t value_count,
                         void *stream) {
    if (!h || ((!in_dev || !out_dev) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_in != value_count * h->par.dev.L || len_out != value_count * words_per_value<W>(*h)) {
        set_last_error("compose: multi_residues must hold moduli_count*value_count words and the output "
                       "value_count*big_uint_value_len words");
        return PFHE_ERR_BAD_LENGTH;
    }
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return rns_compose_dev(h->par, (const DevWord<W> *)in_dev, (DevWord<W> *)out_dev, value_count, (hipStream_t)stream);
}

template <class W>
int rns_compose_host_impl(const RnsHost *h, const W *in, size_t len_in, W *out, size_t len_out, size_t value_count) {
    if (!h || ((!in || !out) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_in != value_count * h->par.dev.L || len_out != value_count * words_per_value<W>(*h)) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(h->device);  // pooled staging context: no allocation in steady state
    if (!st.ok()) return PFHE_ERR_HIP;
    void *i = nullptr, *o = nullptr;
    PFHE_TRY(st.upload(in, len_in * sizeof(W), &i));
    PFHE_TRY(st.alloc(len_out * sizeof(W), &o));
    PFHE_TRY(rns_compose_dev(h->par, (const DevWord<W> *)i, (DevWord<W> *)o, value_count, st.stream()));
    PFHE_TRY(st.download(out, o, len_out * sizeof(W)));
    return st.finish();
}

inline int small_modulus_check(const RnsHost &h, u64 small_value_modulus) {
    for (u64 q : h.moduli) {
        if (small_value_modulus >= q || small_value_modulus < 2) {  // base.rs:288-292, :337-341
            set_last_error("small_value_modulus must be >= 2 and smaller than every RNS modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
    }
    return PFHE_OK;
}

template <class W>
int rns_wrapping_dev_impl(const RnsHost *h, const W *small_dev, size_t value_count, W *multi_dev, size_t len_out,
                          u64 small_value_modulus, void *stream) {
    if (!h || ((!small_dev || !multi_dev) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_out != value_count * h->par.dev.L) return PFHE_ERR_BAD_LENGTH;
    PFHE_TRY(small_modulus_check(*h, small_value_modulus));
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return rns_wrapping_decompose_dev(h->par, (const DevWord<W> *)small_dev, (DevWord<W> *)multi_dev, value_count,
                                      small_value_modulus, (hipStream_t)stream);
}

template <class W>
int rns_wrapping_host_impl(const RnsHost *h, const W *small, size_t value_count, W *multi, size_t len_out,
                           u64 small_value_modulus) {
    if (!h || ((!small || !multi) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_out != value_count * h->par.dev.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(h->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *i = nullptr, *o = nullptr;
    PFHE_TRY(st.upload(small, value_count * sizeof(W), &i));
    PFHE_TRY(st.alloc(len_out * sizeof(W), &o));
    PFHE_TRY(rns_wrapping_dev_impl<W>(h, (const W *)i, value_count, (W *)o, len_out, small_value_modulus, st.stream()));
    PFHE_TRY(st.download(multi, o, len_out * sizeof(W)));
    return st.finish();
}

// `factors`: L ShoupFactor<W> (value, quotient) pairs.  64-bit pairs are used as given; for 32-bit ones the 64-bit
// quotient the kernels multiply by is derived from the value (the product is the same canonical residue)
template <class W>
int rns_add_scaled_dev_impl(const RnsHost *h, const W *small_dev, size_t value_count, W *acc_dev, size_t len_acc,
                            u64 small_value_modulus, bool centred, const W *factors, void *stream) {
    if (!h || !factors || ((!small_dev || !acc_dev) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    const u32 L = h->par.dev.L;
    if (len_acc != value_count * L) return PFHE_ERR_BAD_LENGTH;
    if (centred) PFHE_TRY(small_modulus_check(*h, small_value_modulus));
    std::vector<u64> pairs(2 * (size_t)L);
    for (u32 i = 0; i < L; ++i) {
        if (factors[2 * i] >= h->moduli[i]) {
            set_last_error("factor values must be reduced modulo their modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        pairs[2 * i] = factors[2 * i];
        pairs[2 * i + 1] = sizeof(W) == 8 ? (u64)factors[2 * i + 1]
                                          : (u64)((((unsigned __int128)factors[2 * i]) << 64) / h->moduli[i]);
    }
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    // base.rs:343,371-378: a small modulus of two takes the unsigned branch (a 1 stays +1)
    const bool lift = centred && small_value_modulus != 2;
    return rns_add_decompose_scaled_dev(h->par, (const DevWord<W> *)small_dev, (DevWord<W> *)acc_dev, value_count,
                                        small_value_modulus, lift, pairs.data(), (hipStream_t)stream);
}

template <class W>
int rns_add_scaled_host_impl(const RnsHost *h, const W *small, size_t value_count, W *acc, size_t len_acc,
                             u64 small_value_modulus, bool centred, const W *factors) {
    if (!h || !factors || ((!small || !acc) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_acc != value_count * h->par.dev.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(h->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *i = nullptr, *a = nullptr;
    PFHE_TRY(st.upload(small, value_count * sizeof(W), &i));
    PFHE_TRY(st.upload(acc, len_acc * sizeof(W), &a));
    PFHE_TRY(rns_add_scaled_dev_impl<W>(h, (const W *)i, value_count, (W *)a, len_acc, small_value_modulus, centred, factors,
                                        st.stream()));
    PFHE_TRY(st.download(acc, a, len_acc * sizeof(W)));
    return st.finish();
}

template <class W>
int rns_decompose_big_dev_impl(const RnsHost *h, const W *values_dev, size_t len_in, W *multi_dev, size_t len_out,
                               size_t value_count, void *stream) {
    if (!h || ((!values_dev || !multi_dev) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_in != value_count * words_per_value<W>(*h) || len_out != value_count * h->par.dev.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return rns_decompose_big_dev(h->par, (const DevWord<W> *)values_dev, (DevWord<W> *)multi_dev, value_count,
                                 (hipStream_t)stream);
}

template <class W>
int rns_decompose_big_host_impl(const RnsHost *h, const W *values, size_t len_in, W *multi, size_t len_out,
                                size_t value_count) {
    if (!h || ((!values || !multi) && value_count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len_in != value_count * words_per_value<W>(*h) || len_out != value_count * h->par.dev.L) return PFHE_ERR_BAD_LENGTH;
    if (value_count == 0) return PFHE_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(h->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *i = nullptr, *o = nullptr;
    PFHE_TRY(st.upload(values, len_in * sizeof(W), &i));
    PFHE_TRY(st.alloc(len_out * sizeof(W), &o));
    PFHE_TRY(rns_decompose_big_dev_impl<W>(h, (const W *)i, len_in, (W *)o, len_out, value_count, st.stream()));
    PFHE_TRY(st.download(multi, o, len_out * sizeof(W)));
    return st.finish();
}

/* ---- BigUintApproxSignedBasis<W> ---- */

inline int basis_create_impl(const RnsHost *rns, uint32_t log_basis, size_t reverse_length, BasisHost &b) {
    if (!rns) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(build_basis(*rns, log_basis, reverse_length, b));
    if (b.par.wide()) {
        DeviceGuard g(b.device);
        if (!g.ok) return PFHE_ERR_NO_DEVICE;
        PFHE_TRY(upload_basis_wide(b));
    }
    return PFHE_OK;
}

template <class W>
int basis_scalars_impl(const BasisHost *b, W *out, size_t len) {
    if (!b || !out) return PFHE_ERR_BAD_ARGUMENT;
    const size_t vw = words_per_value<W>(*b), vl = b->par.dev.value_len, ell = b->par.dev.ell;
    if (len != ell * vw) return PFHE_ERR_BAD_LENGTH;
    for (size_t j = 0; j < ell; ++j)
        for (size_t w = 0; w < vw; ++w) out[j * vw + w] = word_of<W>(b->scalars, j * vl, w);
    return PFHE_OK;
}

template <class W>
int basis_scalars_residue_impl(const BasisHost *b, W *out, size_t len) {
    if (!b || !out) return PFHE_ERR_BAD_ARGUMENT;
    if (len != b->scalars_residue.size()) return PFHE_ERR_BAD_LENGTH;
    for (size_t i = 0; i < len; ++i) out[i] = (W)b->scalars_residue[i];
    return PFHE_OK;
}

template <class W>
int basis_init_dev_impl(const BasisHost *b, W *values_dev, size_t len, uint8_t *carries_dev, size_t count, void *stream) {
    if (!b || ((!values_dev || !carries_dev) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;  // basis.rs:332
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return basis_init_value_carry_dev(b->par, (DevWord<W> *)values_dev, carries_dev, count, (hipStream_t)stream);
}

template <class W>
int basis_init_host_impl(const BasisHost *b, W *values, size_t len, uint8_t *carries, size_t count) {
    if (!b || ((!values || !carries) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;
    if (count == 0) return PFHE_OK;
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(b->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *v = nullptr, *c = nullptr;
    PFHE_TRY(st.upload(values, len * sizeof(W), &v));
    PFHE_TRY(st.alloc(count, &c));
    PFHE_TRY(basis_init_value_carry_dev(b->par, (DevWord<W> *)v, (unsigned char *)c, count, st.stream()));
    PFHE_TRY(st.download(values, v, len * sizeof(W)));
    PFHE_TRY(st.download(carries, c, count));
    return st.finish();
}

template <class W>
int basis_init_to_dev_impl(const BasisHost *b, const W *values_dev, size_t len, W *adjusted_dev, uint8_t *carries_dev,
                           size_t count, void *stream) {
    if (!b || ((!values_dev || !adjusted_dev || !carries_dev) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;  // basis.rs:378-379
    if (count == 0) return PFHE_OK;
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    if (adjusted_dev != values_dev)
        PFHE_HIP(hipMemcpyAsync(adjusted_dev, values_dev, len * sizeof(W), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return basis_init_value_carry_dev(b->par, (DevWord<W> *)adjusted_dev, carries_dev, count, (hipStream_t)stream);
}

template <class W>
int basis_init_to_host_impl(const BasisHost *b, const W *values, size_t len, W *adjusted, uint8_t *carries, size_t count) {
    if (!b || ((!values || !adjusted || !carries) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;
    if (count == 0) return PFHE_OK;
    if (adjusted != values) std::memcpy(adjusted, values, len * sizeof(W));
    return basis_init_host_impl<W>(b, adjusted, len, carries, count);
}

template <class W>
int basis_unsigned_dev_impl(const BasisHost *b, size_t level, const W *values_dev, size_t len, W *digits_dev,
                            uint8_t *carries_dev, size_t count, void *stream) {
    if (!b || ((!values_dev || !digits_dev || !carries_dev) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (level >= b->par.dev.ell) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;  // common.rs:316-317
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return basis_unsigned_decompose_dev(b->par, (u32)level, (const DevWord<W> *)values_dev, (DevWord<W> *)digits_dev,
                                        carries_dev, count, (hipStream_t)stream);
}

template <class W>
int basis_unsigned_host_impl(const BasisHost *b, size_t level, const W *values, size_t len, W *digits, uint8_t *carries,
                             size_t count) {
    if (!b || ((!values || !digits || !carries) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (level >= b->par.dev.ell) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b)) return PFHE_ERR_BAD_LENGTH;
    if (count == 0) return PFHE_OK;
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(b->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *v = nullptr, *d = nullptr, *c = nullptr;
    PFHE_TRY(st.upload(values, len * sizeof(W), &v));
    PFHE_TRY(st.alloc(count * sizeof(W), &d));
    PFHE_TRY(st.upload(carries, count, &c));
    PFHE_TRY(basis_unsigned_decompose_dev(b->par, (u32)level, (const DevWord<W> *)v, (DevWord<W> *)d, (unsigned char *)c, count,
                                          st.stream()));
    PFHE_TRY(st.download(digits, d, count * sizeof(W)));
    PFHE_TRY(st.download(carries, c, count));
    return st.finish();
}

template <class W>
int basis_signed_dev_impl(const BasisHost *b, size_t level, const W *values_dev, size_t len, W *decomposed_dev,
                          size_t len_out, uint8_t *carries_dev, size_t count, void *stream) {
    if (!b || ((!values_dev || !decomposed_dev || !carries_dev) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (level >= b->par.dev.ell) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b) || len_out != len) return PFHE_ERR_BAD_LENGTH;  // common.rs:296-297
    if (count && values_dev == decomposed_dev) {
        set_last_error("decompose_slice_to needs distinct input and output buffers");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return basis_signed_decompose_dev(b->rns, b->par, (u32)level, (const DevWord<W> *)values_dev,
                                      (DevWord<W> *)decomposed_dev, carries_dev, count, (hipStream_t)stream);
}

template <class W>
int basis_signed_host_impl(const BasisHost *b, size_t level, const W *values, size_t len, W *decomposed, size_t len_out,
                           uint8_t *carries, size_t count) {
    if (!b || ((!values || !decomposed || !carries) && count)) return PFHE_ERR_BAD_ARGUMENT;
    if (level >= b->par.dev.ell) return PFHE_ERR_BAD_ARGUMENT;
    if (len != count * words_per_value<W>(*b) || len_out != len) return PFHE_ERR_BAD_LENGTH;
    if (count == 0) return PFHE_OK;
    DeviceGuard g(b->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(b->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *v = nullptr, *d = nullptr, *c = nullptr;
    PFHE_TRY(st.upload(values, len * sizeof(W), &v));
    PFHE_TRY(st.alloc(len * sizeof(W), &d));
    PFHE_TRY(st.upload(carries, count, &c));
    PFHE_TRY(basis_signed_decompose_dev(b->rns, b->par, (u32)level, (const DevWord<W> *)v, (DevWord<W> *)d,
                                        (unsigned char *)c, count, st.stream()));
    PFHE_TRY(st.download(decomposed, d, len * sizeof(W)));
    PFHE_TRY(st.download(carries, c, count));
    return st.finish();
}

}  // namespace

// the entry points of one word width: NS = pfhe_rns / pfhe_rns32, BS = pfhe_basis / pfhe_basis32, W = the C word type
#define H(p) ((p) ? &(p)->h : nullptr)
#define PFHE_RNS_FAMILY(NS, BS, W)                                                                                         \
    int NS##_create(const W *moduli, size_t count, int device, NS **out) {                                                \
        PFHE_GUARD_BEGIN                                                                                                   \
        if (!out) return PFHE_ERR_BAD_ARGUMENT;                                                                            \
        *out = nullptr;                                                                                                    \
        auto r = std::make_unique<NS>();                                                                                   \
        PFHE_TRY(rns_create_impl<W>(moduli, count, device, r->h));                                                         \
        *out = r.release();                                                                                                \
        return PFHE_OK;                                                                                                    \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    void NS##_destroy(NS *r) { delete r; }                                                                                 \
    size_t NS##_moduli_count(const NS *r) { return r ? r->h.par.dev.L : 0; }                                               \
    size_t NS##_big_uint_value_len(const NS *r) { return r ? words_per_value<W>(r->h) : 0; }                               \
    int NS##_moduli_product(const NS *r, W *out, size_t len) { return rns_moduli_product_impl<W>(H(r), out, len); }        \
    int NS##_compose_multiple_values_to_dev(const NS *r, const W *multi_residues_dev, size_t len_in,                      \
                                            W *big_uint_values_dev, size_t len_out, size_t value_count, void *stream) {   \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_compose_dev_impl<W>(H(r), multi_residues_dev, len_in, big_uint_values_dev, len_out, value_count, stream); \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_compose_multiple_values_to(const NS *r, const W *multi_residues, size_t len_in, W *big_uint_values,          \
                                        size_t len_out, size_t value_count) {                                              \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_compose_host_impl<W>(H(r), multi_residues, len_in, big_uint_values, len_out, value_count);              \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_wrapping_decompose_small_values_to_dev(const NS *r, const W *small_values_dev, size_t value_count,           \
                                                    W *multi_residues_dev, size_t len_out, W small_value_modulus,         \
                                                    void *stream) {                                                        \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_wrapping_dev_impl<W>(H(r), small_values_dev, value_count, multi_residues_dev, len_out,                  \
                                        small_value_modulus, stream);                                                      \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_wrapping_decompose_small_values_to(const NS *r, const W *small_values, size_t value_count,                   \
                                                W *multi_residues, size_t len_out, W small_value_modulus) {               \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_wrapping_host_impl<W>(H(r), small_values, value_count, multi_residues, len_out, small_value_modulus);   \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_add_wrapping_decompose_small_values_scaled_dev(const NS *r, const W *small_values_dev, size_t value_count,   \
                                                            W *acc_dev, size_t len_acc, W small_value_modulus,            \
                                                            const W *factors, void *stream) {                              \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_add_scaled_dev_impl<W>(H(r), small_values_dev, value_count, acc_dev, len_acc, small_value_modulus,      \
                                          true, factors, stream);                                                          \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_add_decompose_small_values_scaled_dev(const NS *r, const W *small_values_dev, size_t value_count,            \
                                                   W *acc_dev, size_t len_acc, const W *factors, void *stream) {           \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_add_scaled_dev_impl<W>(H(r), small_values_dev, value_count, acc_dev, len_acc, 0, false, factors, stream); \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_add_wrapping_decompose_small_values_scaled(const NS *r, const W *small_values, size_t value_count, W *acc,   \
                                                        size_t len_acc, W small_value_modulus, const W *factors) {         \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_add_scaled_host_impl<W>(H(r), small_values, value_count, acc, len_acc, small_value_modulus, true, factors); \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_add_decompose_small_values_scaled(const NS *r, const W *small_values, size_t value_count, W *acc,            \
                                               size_t len_acc, const W *factors) {                                         \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_add_scaled_host_impl<W>(H(r), small_values, value_count, acc, len_acc, 0, false, factors);              \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_decompose_big_uint_values_to_dev(const NS *r, const W *big_uint_values_dev, size_t len_in,                   \
                                              W *multi_residues_dev, size_t len_out, size_t value_count, void *stream) {  \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_decompose_big_dev_impl<W>(H(r), big_uint_values_dev, len_in, multi_residues_dev, len_out, value_count,  \
                                             stream);                                                                      \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int NS##_decompose_big_uint_values_to(const NS *r, const W *big_uint_values, size_t len_in, W *multi_residues,        \
                                          size_t len_out, size_t value_count) {                                            \
        PFHE_GUARD_BEGIN                                                                                                   \
        return rns_decompose_big_host_impl<W>(H(r), big_uint_values, len_in, multi_residues, len_out, value_count);        \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_create(const NS *rns, uint32_t log_basis, size_t reverse_length, BS **out) {                                 \
        PFHE_GUARD_BEGIN                                                                                                   \
        if (!out || !rns) return PFHE_ERR_BAD_ARGUMENT;                                                                    \
        *out = nullptr;                                                                                                    \
        auto b = std::make_unique<BS>();                                                                                   \
        PFHE_TRY(basis_create_impl(H(rns), log_basis, reverse_length, b->h));                                              \
        *out = b.release();                                                                                                \
        return PFHE_OK;                                                                                                    \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    void BS##_destroy(BS *b) { delete b; }                                                                                 \
    size_t BS##_decompose_length(const BS *b) { return b ? b->h.par.dev.ell : 0; }                                         \
    uint32_t BS##_log_basis(const BS *b) { return b ? b->h.par.dev.log_basis : 0; }                                        \
    uint32_t BS##_drop_bits(const BS *b) { return b ? b->h.par.dev.drop_bits : 0; }                                        \
    W BS##_basis_value(const BS *b) { return b ? (W)b->h.par.dev.basis : 0; }                                              \
    int BS##_scalars(const BS *b, W *out, size_t len) { return basis_scalars_impl<W>(H(b), out, len); }                    \
    int BS##_scalars_residue(const BS *b, W *out, size_t len) { return basis_scalars_residue_impl<W>(H(b), out, len); }    \
    int BS##_init_value_carry_slice_inplace_dev(const BS *b, W *values_dev, size_t len, uint8_t *carries_dev,             \
                                                size_t count, void *stream) {                                              \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_init_dev_impl<W>(H(b), values_dev, len, carries_dev, count, stream);                                  \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_init_value_carry_slice_inplace(const BS *b, W *values, size_t len, uint8_t *carries, size_t count) {         \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_init_host_impl<W>(H(b), values, len, carries, count);                                                 \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_unsigned_decompose_slice_to_dev(const BS *b, size_t level, const W *values_dev, size_t len, W *digits_dev,   \
                                             uint8_t *carries_dev, size_t count, void *stream) {                           \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_unsigned_dev_impl<W>(H(b), level, values_dev, len, digits_dev, carries_dev, count, stream);           \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_unsigned_decompose_slice_to(const BS *b, size_t level, const W *values, size_t len, W *digits,               \
                                         uint8_t *carries, size_t count) {                                                 \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_unsigned_host_impl<W>(H(b), level, values, len, digits, carries, count);                              \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_init_value_carry_slice_to_dev(const BS *b, const W *values_dev, size_t len, W *adjusted_dev,                 \
                                           uint8_t *carries_dev, size_t count, void *stream) {                             \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_init_to_dev_impl<W>(H(b), values_dev, len, adjusted_dev, carries_dev, count, stream);                 \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_init_value_carry_slice_to(const BS *b, const W *values, size_t len, W *adjusted, uint8_t *carries,           \
                                       size_t count) {                                                                     \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_init_to_host_impl<W>(H(b), values, len, adjusted, carries, count);                                    \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_decompose_slice_to_dev(const BS *b, size_t level, const W *values_dev, size_t len, W *decomposed_dev,        \
                                    size_t len_out, uint8_t *carries_dev, size_t count, void *stream) {                    \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_signed_dev_impl<W>(H(b), level, values_dev, len, decomposed_dev, len_out, carries_dev, count, stream); \
        PFHE_GUARD_END                                                                                                     \
    }                                                                                                                      \
    int BS##_decompose_slice_to(const BS *b, size_t level, const W *values, size_t len, W *decomposed, size_t len_out,    \
                                uint8_t *carries, size_t count) {                                                          \
        PFHE_GUARD_BEGIN                                                                                                   \
        return basis_signed_host_impl<W>(H(b), level, values, len, decomposed, len_out, carries, count);                   \
        PFHE_GUARD_END                                                                                                     \
    }

extern "C" {

PFHE_RNS_FAMILY(pfhe_rns, pfhe_basis, uint64_t)
PFHE_RNS_FAMILY(pfhe_rns32, pfhe_basis32, uint32_t)

/* ------------------------------ external product ------------------------------ */

int pfhe_extprod_plan_create(const pfhe_dcrt *table, const pfhe_rns *rns, const pfhe_basis *basis,
                             size_t glwe_dimension, size_t chunk, pfhe_extprod_plan **out) {
    PFHE_GUARD_BEGIN
    if (!out || !table || !rns || !basis || glwe_dimension == 0 || glwe_dimension > 64) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    const TableSet *t = capi_table_of(table);
    if (t->L != rns->h.par.dev.L) {
        set_last_error("DCRT table and RNS base have different moduli counts");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    for (u32 i = 0; i < t->L; ++i) {
        if (t->primes[i].q != rns->h.moduli[i]) {
            set_last_error("DCRT table and RNS base must use the same moduli in the same order");
            return PFHE_ERR_BAD_ARGUMENT;
        }
    }
    if (basis->h.rns.dev.L != rns->h.par.dev.L || basis->h.Q != rns->h.Q) return PFHE_ERR_BAD_ARGUMENT;  // basis.rs:52
    for (u32 i = 0; i < t->L; ++i) {
        if (basis->h.par.dev.basis >= t->primes[i].q) {  // wrapping_decompose needs B < q_i (base.rs:288-292)
            set_last_error("gadget basis must be smaller than every RNS modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
    }
    auto p = std::make_unique<pfhe_extprod_plan>();
    p->table = t;
    p->rns = rns->h.par;
    p->basis_par = basis->h.par;
    p->basis = basis->h.par.dev;
    p->k = (u32)glwe_dimension;
    // default: about 2 GiB of digit polynomials per buffer, at least 128 ciphertexts — measured at N = 2^16, 3 limbs
    // with today's kernels (ms per 1024 products): 32: 21.7, 64: 21.2, 128: 20.8, 256: 20.9, 1024: 20.8 (round 1, slower
    // kernels: 64 was the optimum); small rings need many more ciphertexts per launch to amortise the launches
    if (chunk == 0) {
        const size_t per_ct = (size_t)(glwe_dimension + 1) * p->basis.ell * t->L * t->n * sizeof(u64);
        chunk = std::max<size_t>(128, std::min<size_t>(65536, ((size_t)2 << 30) / per_ct));
    }
    p->chunk = chunk;
    p->digits_words = p->chunk * (p->k + 1) * p->basis.ell * t->L * t->n;
    DeviceGuard g(t->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    // read here, once, and kept in the plan (nothing on the launch path calls getenv)
    p->use_fused = std::getenv("PFHE_DISABLE_FUSED_EXTPROD") == nullptr;
    {
        void *d = nullptr;
        PFHE_HIP(counted_malloc(&d, p->digits_words * sizeof(u64)));
        p->digits = (u64 *)d;
    }
    if (gadget_decompose_strided_supported(t->log_n, p->rns.dev.value_len) ||
        extprod_small_supported(t->log_n, p->k, p->rns.dev.value_len, p->basis.log_basis)) {
        void *d = nullptr;
        p->sdigit_bytes = gadget_digit_bytes(p->basis.log_basis);
        PFHE_HIP(counted_malloc(&d, p->chunk * (p->k + 1) * p->basis.ell * t->n * p->sdigit_bytes));
        p->sdigits = d;
    }
    PFHE_HIP(hipEventCreateWithFlags(&p->last_done, hipEventDisableTiming));
    *out = p.release();
    return PFHE_OK;
    PFHE_GUARD_END
}

void pfhe_extprod_plan_destroy(pfhe_extprod_plan *p) { delete p; }
int pfhe_extprod_plan_in_use(const pfhe_extprod_plan *p) {
    return p && p->owner.load(std::memory_order_acquire) != 0 ? 1 : 0;
}
// test aid: hold != 0 takes the plan for the calling thread as an entry point would (PFHE_ERR_BUSY if another thread has
// it) and keeps it until the same thread calls with hold == 0
int pfhe_extprod_plan_debug_hold(pfhe_extprod_plan *p, int hold) {
    if (plan_check(p) != PFHE_OK) return PFHE_ERR_BAD_ARGUMENT;
    if (hold) return plan_acquire(p) ? PFHE_OK : PFHE_ERR_BUSY;
    if (p->owner.load(std::memory_order_relaxed) != plan_thread_token()) return PFHE_ERR_BAD_ARGUMENT;
    plan_release(p);
    return PFHE_OK;
}
size_t pfhe_extprod_plan_scratch_bytes(const pfhe_extprod_plan *p) {
    if (!p) return 0;
    return p->digits_words * 8 + (p->sdigits ? p->chunk * (p->k + 1) * p->basis.ell * p->table->n * p->sdigit_bytes : 0);
}

int pfhe_extprod_mul_dcrt_ggsw_to_dev(pfhe_extprod_plan *plan, const uint64_t *crt_glwe_dev, size_t len_glwe,
                                      const uint64_t *dcrt_ggsw_dev, size_t len_ggsw, uint64_t *result_dev,
                                      size_t len_result, int into_coeff_form, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W;
    const size_t ggsw = (size_t)(plan->k + 1) * plan->basis.ell * glwe;
    if (len_glwe % glwe != 0 || len_result != len_glwe || (len_ggsw != ggsw && len_ggsw != len_glwe / glwe * ggsw)) {
        set_last_error("external product: glwe/result must be batch*(k+1)*L*N words and the GGSW one or batch "
                       "ciphertexts of (k+1)*ell*(k+1)*L*N words");
        return PFHE_ERR_BAD_LENGTH;
    }
    const u64 batch = len_glwe / glwe;
    if (batch == 0) return PFHE_OK;
    if (!crt_glwe_dev || !dcrt_ggsw_dev || !result_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(crt_glwe_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_ggsw_dev);
    PFHE_REQUIRE_ALIGNED(result_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    const bool shared = len_ggsw == ggsw && batch > 1 ? true : (len_ggsw == ggsw);
    // result.set_zero() (glwe/crt.rs:217) is implied: the first accumulation overwrites
    int coeff_passes = 0;
    PFHE_TRY(run_product(plan, (const u64 *)crt_glwe_dev, plan->k + 1, (const u64 *)dcrt_ggsw_dev, shared,
                         (u64 *)result_dev, batch, false, (hipStream_t)stream, into_coeff_form != 0, &coeff_passes));
    if (into_coeff_form && coeff_passes == 0)  // DcrtGlwe::into_coeff_form, macros/mod.rs:901-911
        PFHE_TRY(ntt_inverse_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, (u64 *)result_dev, batch * (plan->k + 1) * t.L, false,
                                 (hipStream_t)stream, t.tune));
    for (int i = coeff_passes; into_coeff_form && i > 0 && i < ntt_num_passes(t.log_n, t.ntt_arith, t.tune); ++i)
        PFHE_TRY(ntt_pass_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, (u64 *)result_dev, batch * (plan->k + 1) * t.L, true, i,
                              false, (hipStream_t)stream, nullptr, 0, t.tune));
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_extprod_profile_dev(pfhe_extprod_plan *plan, const uint64_t *crt_glwe_dev, size_t len_glwe,
                             const uint64_t *dcrt_ggsw_dev, size_t len_ggsw, uint64_t *result_dev, size_t len_result,
                             double *ms_out, size_t *launches_out, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    if (!ms_out || !launches_out) return PFHE_ERR_BAD_ARGUMENT;
    std::vector<hipEvent_t> ev;
    plan->prof = &ev;
    // coefficient-form output, as bench.py times it: the second group includes the inverse block pass fused into the
    // multiply-accumulate kernel; the final strided inverse pass runs after the last stamp
    int rc = pfhe_extprod_mul_dcrt_ggsw_to_dev(plan, crt_glwe_dev, len_glwe, dcrt_ggsw_dev, len_ggsw, result_dev,
                                               len_result, 1, stream);
    plan->prof = nullptr;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    ms_out[0] = ms_out[1] = 0.0;
    *launches_out = ev.size() / 3;
    for (size_t i = 0; rc == PFHE_OK && e == hipSuccess && i + 2 < ev.size(); i += 3) {
        float a = 0, b = 0;
        e = hipEventElapsedTime(&a, ev[i], ev[i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, ev[i + 1], ev[i + 2]);
        ms_out[0] += a;  // digit extraction + lifting strided pass
        ms_out[1] += b;  // block pass of the transform + multiply-accumulate
    }
    for (hipEvent_t x : ev) (void)hipEventDestroy(x);
    if (e != hipSuccess) return hip_fail(e, "external-product profile", __FILE__, __LINE__);
    return rc;
    PFHE_GUARD_END
}

int pfhe_extprod_add_dcrt_glev_mul_crt_poly_assign_dev(pfhe_extprod_plan *plan, uint64_t *acc_dev, size_t len_acc,
                                                       const uint64_t *dcrt_glev_dev, size_t len_glev,
                                                       const uint64_t *crt_poly_dev, size_t len_poly, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W, glev = plan->basis.ell * glwe;
    if (len_poly % W != 0) return PFHE_ERR_BAD_LENGTH;
    const u64 batch = len_poly / W;
    if (len_acc != batch * glwe || (len_glev != glev && len_glev != batch * glev)) {
        set_last_error("glev product: acc must be batch*(k+1)*L*N words and the GLev one or batch of ell*(k+1)*L*N");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (batch == 0) return PFHE_OK;
    if (!acc_dev || !dcrt_glev_dev || !crt_poly_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(acc_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glev_dev);
    PFHE_REQUIRE_ALIGNED(crt_poly_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return run_product(plan, (const u64 *)crt_poly_dev, 1, (const u64 *)dcrt_glev_dev, len_glev == glev,
                       (u64 *)acc_dev, batch, true, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_extprod_glev_mul_crt_poly_to_dev(pfhe_extprod_plan *plan, const uint64_t *dcrt_glev_dev, size_t len_glev,
                                          const uint64_t *crt_poly_dev, size_t len_poly, uint64_t *result_dev,
                                          size_t len_result, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W, glev = plan->basis.ell * glwe;
    if (len_poly % W != 0) return PFHE_ERR_BAD_LENGTH;
    const u64 batch = len_poly / W;
    if (len_result != batch * glwe || (len_glev != glev && len_glev != batch * glev)) {
        set_last_error("glev product: result must be batch*(k+1)*L*N words and the GLev one or batch of ell*(k+1)*L*N");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (batch == 0) return PFHE_OK;
    if (!result_dev || !dcrt_glev_dev || !crt_poly_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(result_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glev_dev);
    PFHE_REQUIRE_ALIGNED(crt_poly_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return run_product(plan, (const u64 *)crt_poly_dev, 1, (const u64 *)dcrt_glev_dev, len_glev == glev,
                       (u64 *)result_dev, batch, false, (hipStream_t)stream);
    PFHE_GUARD_END
}

// DcrtGlwe::add_dcrt_glev_mul_big_uint_poly_assign (glwe/dcrt.rs:258-338) / DcrtGlev::mul_big_uint_poly_to
// (glev/dcrt.rs:113-175): the GLev rows against polynomials given as big integers modulo Q
static int glev_big_uint_common(pfhe_extprod_plan *plan, uint64_t *out_dev, size_t len_out, const uint64_t *dcrt_glev_dev,
                                size_t len_glev, const uint64_t *big_uint_poly_dev, size_t len_poly, bool accumulate,
                                void *stream) {
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W, glev = plan->basis.ell * glwe;
    const size_t in_words = (size_t)plan->rns.dev.value_len * t.n;
    if (len_poly % in_words != 0) return PFHE_ERR_BAD_LENGTH;  // glwe/dcrt.rs:277
    const u64 batch = len_poly / in_words;
    if (len_out != batch * glwe || (len_glev != glev && len_glev != batch * glev)) {
        set_last_error("glev product: acc/result must be batch*(k+1)*L*N words and the GLev one or batch of ell*(k+1)*L*N");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (batch == 0) return PFHE_OK;
    if (!out_dev || !dcrt_glev_dev || !big_uint_poly_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(out_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glev_dev);
    PFHE_REQUIRE_ALIGNED(big_uint_poly_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return run_product(plan, (const u64 *)big_uint_poly_dev, 1, (const u64 *)dcrt_glev_dev, len_glev == glev,
                       (u64 *)out_dev, batch, accumulate, (hipStream_t)stream, false, nullptr, true);
}

int pfhe_extprod_add_dcrt_glev_mul_big_uint_poly_assign_dev(pfhe_extprod_plan *plan, uint64_t *acc_dev, size_t len_acc,
                                                            const uint64_t *dcrt_glev_dev, size_t len_glev,
                                                            const uint64_t *big_uint_poly_dev, size_t len_poly,
                                                            void *stream) {
    PFHE_GUARD_BEGIN
    return glev_big_uint_common(plan, acc_dev, len_acc, dcrt_glev_dev, len_glev, big_uint_poly_dev, len_poly, true, stream);
    PFHE_GUARD_END
}

int pfhe_extprod_glev_mul_big_uint_poly_to_dev(pfhe_extprod_plan *plan, const uint64_t *dcrt_glev_dev, size_t len_glev,
                                               const uint64_t *big_uint_poly_dev, size_t len_poly, uint64_t *result_dev,
                                               size_t len_result, void *stream) {
    PFHE_GUARD_BEGIN
    return glev_big_uint_common(plan, result_dev, len_result, dcrt_glev_dev, len_glev, big_uint_poly_dev, len_poly, false,
                                stream);
    PFHE_GUARD_END
}

int pfhe_extprod_mul_dcrt_ggsw_to(pfhe_extprod_plan *plan, const uint64_t *crt_glwe, size_t len_glwe,
                                  const uint64_t *dcrt_ggsw, size_t len_ggsw, uint64_t *result, size_t len_result,
                                  int into_coeff_form) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan_check(plan));
    PFHE_PLAN_LEASE(plan);
    if ((!crt_glwe || !dcrt_ggsw || !result) && len_glwe) return PFHE_ERR_BAD_ARGUMENT;
    DeviceGuard g(plan->table->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(plan->table->device);  // pooled staging context: no allocation in steady state
    if (!st.ok()) return PFHE_ERR_HIP;
    void *a = nullptr, *k = nullptr, *r = nullptr;
    PFHE_TRY(st.upload(crt_glwe, len_glwe * 8, &a));
    PFHE_TRY(st.upload(dcrt_ggsw, len_ggsw * 8, &k));
    PFHE_TRY(st.alloc(len_result * 8, &r));
    PFHE_TRY(pfhe_extprod_mul_dcrt_ggsw_to_dev(plan, (const uint64_t *)a, len_glwe, (const uint64_t *)k, len_ggsw,
                                               (uint64_t *)r, len_result, into_coeff_form, st.stream()));
    PFHE_TRY(st.download(result, r, len_result * 8));
    return st.finish();
    PFHE_GUARD_END
}

/* ------------------------------ external product over U32DcrtTable ------------------------------ */

}  // extern "C"

namespace {

// *coeff_passes (when the caller wants coefficient form): 1 = the inverse transform's block pass ran inside the fused
// kernel (the caller runs the strided pass), 0 = none
int run_product32(pfhe_extprod32_plan *p, const u32 *polys, u32 rows, const u32 *keys, bool keys_shared, u32 *result,
                  u64 batch, bool accumulate, hipStream_t s, bool big_input, bool into_coeff = false,
                  int *coeff_passes = nullptr) {
    if (coeff_passes) *coeff_passes = 0;
    const TableSet &t = *p->table;
    const bool tracked = p->last_done != nullptr && !stream_is_capturing(s);
    if (tracked && p->last_valid) PFHE_HIP(hipStreamWaitEvent(s, p->last_done, 0));
    const u64 W = (u64)t.L * t.n;
    RnsParams rns = p->rns;
    rns.dev.big_input = rns.wide_tab.big_input = big_input ? 1u : 0u;
    const u64 in_words = big_input ? (u64)rns.dev.value_words * t.n : W;
    const u32 ell = p->basis.ell;
    const u64 key_words = (u64)rows * ell * (p->k + 1) * W;
    // the fused kernels launch one workgroup per (ciphertext, limb, block): they pay once that fills the chip
    const bool fused = p->sdigits != nullptr && p->use_fused && extprod32_fused_supported(t.log_n, p->k) &&
                       ((std::min<u64>(batch, p->chunk) * t.L) << (t.log_n - 12)) >= 160;
    const bool inv_tail = fused && into_coeff && !accumulate && coeff_passes != nullptr;
    if (inv_tail) *coeff_passes = 1;
    int rc = PFHE_OK;
    for (u64 done = 0; done < batch && rc == PFHE_OK; done += p->chunk) {
        const u64 cur = std::min<u64>(p->chunk, batch - done);
        const u32 *kp = keys + (keys_shared ? 0 : done * key_words);
        u32 *rp = result + done * (p->k + 1) * W;
        if (fused) {
            rc = gadget_signed_digits_dev<u32>(rns, p->basis_par, t.log_n, polys + done * rows * in_words, p->sdigits, cur * rows, s);
            if (rc == PFHE_OK) rc = digits_strided32_dev(t.primes_dev, t.L, t.log_n, ell, p->sdigits, p->digits, cur * rows, s);
            if (rc == PFHE_OK)
                rc = gadget_block_mulacc32_dev(t.primes_dev, t.L, t.log_n, rows * ell, p->digits, kp, keys_shared, rp, cur,
                                               accumulate, inv_tail, s);
            continue;
        }
        rc = gadget_decompose_dev<u32>(rns, p->basis_par, t.log_n, polys + done * rows * in_words, p->digits, cur * rows, s);
        if (rc == PFHE_OK)
            rc = ntt32_transform_dev(t.primes_dev, t.L, t.log_n, p->digits, cur * rows * ell * t.L, false, false, s, t.tune);
        if (rc == PFHE_OK)
            rc = gadget_mulacc32_dev(t.primes_dev, t.L, t.log_n, p->k, rows, ell, p->digits, kp, keys_shared, rp, cur, accumulate, s);
    }
    if (tracked) {  // also after a failed call: whatever it queued still uses the buffers
        if (hipEventRecord(p->last_done, s) == hipSuccess) {
            p->last_valid = true;
        } else {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);
            p->last_valid = false;
        }
    }
    return rc;
}

int plan32_check(const pfhe_extprod32_plan *p) { return (!p || !p->table) ? PFHE_ERR_BAD_ARGUMENT : PFHE_OK; }

// one GLev row against `batch` polynomials (CRT residues or big integers), accumulating or overwriting
int glev32_common(pfhe_extprod32_plan *plan, uint32_t *out_dev, size_t len_out, const uint32_t *dcrt_glev_dev, size_t len_glev,
                  const uint32_t *poly_dev, size_t len_poly, bool accumulate, bool big_input, void *stream) {
    PFHE_TRY(plan32_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W, glev = plan->basis.ell * glwe;
    const size_t in_words = big_input ? (size_t)plan->rns.dev.value_words * t.n : W;
    if (len_poly % in_words != 0) return PFHE_ERR_BAD_LENGTH;
    const u64 batch = len_poly / in_words;
    if (len_out != batch * glwe || (len_glev != glev && len_glev != batch * glev)) {
        set_last_error("glev product: acc/result must be batch*(k+1)*L*N words and the GLev one or batch of ell*(k+1)*L*N");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (batch == 0) return PFHE_OK;
    if (!out_dev || !dcrt_glev_dev || !poly_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(out_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glev_dev);
    PFHE_REQUIRE_ALIGNED(poly_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return run_product32(plan, poly_dev, 1, dcrt_glev_dev, len_glev == glev, out_dev, batch, accumulate, (hipStream_t)stream,
                         big_input);
}

}  // namespace

extern "C" {

int pfhe_extprod32_plan_create(const pfhe_dcrt32 *table, const pfhe_rns32 *rns, const pfhe_basis32 *basis,
                               size_t glwe_dimension, size_t chunk, pfhe_extprod32_plan **out) {
    PFHE_GUARD_BEGIN
    if (!out || !table || !rns || !basis || glwe_dimension == 0 || glwe_dimension > 64) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    const TableSet *t = capi_table32_of(table);
    if (t->L != rns->h.par.dev.L) {
        set_last_error("DCRT table and RNS base have different moduli counts");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    for (u32 i = 0; i < t->L; ++i) {
        if (t->primes[i].q != rns->h.moduli[i]) {
            set_last_error("DCRT table and RNS base must use the same moduli in the same order");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        if (basis->h.par.dev.basis >= t->primes[i].q) {  // wrapping_decompose needs B < q_i (base.rs:288-292)
            set_last_error("gadget basis must be smaller than every RNS modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
    }
    if (basis->h.rns.dev.L != rns->h.par.dev.L || basis->h.Q != rns->h.Q) return PFHE_ERR_BAD_ARGUMENT;  // basis.rs:52
    auto p = std::make_unique<pfhe_extprod32_plan>();
    p->table = t;
    p->rns = rns->h.par;
    p->basis_par = basis->h.par;
    p->basis = basis->h.par.dev;
    p->k = (u32)glwe_dimension;
    if (chunk == 0) {  // about 1 GiB of digit polynomials, at least 128 ciphertexts
        const size_t per_ct = (size_t)(glwe_dimension + 1) * p->basis.ell * t->L * t->n * sizeof(u32);
        chunk = std::max<size_t>(128, std::min<size_t>(65536, ((size_t)1 << 30) / per_ct));
    }
    p->chunk = chunk;
    p->digits_words = p->chunk * (p->k + 1) * p->basis.ell * t->L * t->n;
    DeviceGuard g(t->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    p->use_fused = std::getenv("PFHE_DISABLE_FUSED_EXTPROD") == nullptr;
    void *d = nullptr;
    PFHE_HIP(counted_malloc(&d, p->digits_words * sizeof(u32)));
    p->digits = (u32 *)d;
    if (p->use_fused && extprod32_fused_supported(t->log_n, p->k)) {
        void *sd = nullptr;
        PFHE_HIP(counted_malloc(&sd, p->chunk * (p->k + 1) * p->basis.ell * t->n * sizeof(int)));
        p->sdigits = (int *)sd;
    }
    PFHE_HIP(hipEventCreateWithFlags(&p->last_done, hipEventDisableTiming));
    *out = p.release();
    return PFHE_OK;
    PFHE_GUARD_END
}

void pfhe_extprod32_plan_destroy(pfhe_extprod32_plan *p) { delete p; }
int pfhe_extprod32_plan_in_use(const pfhe_extprod32_plan *p) {
    return p && p->owner.load(std::memory_order_acquire) != 0 ? 1 : 0;
}
size_t pfhe_extprod32_plan_scratch_bytes(const pfhe_extprod32_plan *p) {
    if (!p) return 0;
    return p->digits_words * sizeof(u32) + (p->sdigits ? p->chunk * (p->k + 1) * p->basis.ell * p->table->n * sizeof(int) : 0);
}

int pfhe_extprod32_mul_dcrt_ggsw_to_dev(pfhe_extprod32_plan *plan, const uint32_t *crt_glwe_dev, size_t len_glwe,
                                        const uint32_t *dcrt_ggsw_dev, size_t len_ggsw, uint32_t *result_dev,
                                        size_t len_result, int into_coeff_form, void *stream) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan32_check(plan));
    PFHE_PLAN_LEASE(plan);
    const TableSet &t = *plan->table;
    const size_t W = (size_t)t.L * t.n, glwe = (plan->k + 1) * W;
    const size_t ggsw = (size_t)(plan->k + 1) * plan->basis.ell * glwe;
    if (len_glwe % glwe != 0 || len_result != len_glwe || (len_ggsw != ggsw && len_ggsw != len_glwe / glwe * ggsw)) {
        set_last_error("external product: glwe/result must be batch*(k+1)*L*N words and the GGSW one or batch "
                       "ciphertexts of (k+1)*ell*(k+1)*L*N words");
        return PFHE_ERR_BAD_LENGTH;
    }
    const u64 batch = len_glwe / glwe;
    if (batch == 0) return PFHE_OK;
    if (!crt_glwe_dev || !dcrt_ggsw_dev || !result_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(crt_glwe_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_ggsw_dev);
    PFHE_REQUIRE_ALIGNED(result_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    // result.set_zero() (glwe/crt.rs:217) is implied: the multiply-accumulate overwrites
    int coeff_passes = 0;
    PFHE_TRY(run_product32(plan, crt_glwe_dev, plan->k + 1, dcrt_ggsw_dev, len_ggsw == ggsw, result_dev, batch, false,
                           (hipStream_t)stream, false, into_coeff_form != 0, &coeff_passes));
    if (into_coeff_form && coeff_passes == 0)  // DcrtGlwe::into_coeff_form, macros/mod.rs:901-911
        PFHE_TRY(ntt32_transform_dev(t.primes_dev, t.L, t.log_n, result_dev, batch * (plan->k + 1) * t.L, true, false,
                                     (hipStream_t)stream, t.tune));
    if (into_coeff_form && coeff_passes == 1)  // the block pass ran inside the fused kernel: the strided pass finishes
        PFHE_TRY(ntt_pass_dev(t.primes_dev, t.L, t.log_n - 1, kArithB32, reinterpret_cast<u64 *>(result_dev),
                              batch * (plan->k + 1) * t.L, true, 1, false, (hipStream_t)stream, nullptr, 0, t.tune));
    return PFHE_OK;
    PFHE_GUARD_END
}

int pfhe_extprod32_mul_dcrt_ggsw_to(pfhe_extprod32_plan *plan, const uint32_t *crt_glwe, size_t len_glwe,
                                    const uint32_t *dcrt_ggsw, size_t len_ggsw, uint32_t *result, size_t len_result,
                                    int into_coeff_form) {
    PFHE_GUARD_BEGIN
    PFHE_TRY(plan32_check(plan));
    PFHE_PLAN_LEASE(plan);
    if ((!crt_glwe || !dcrt_ggsw || !result) && len_glwe) return PFHE_ERR_BAD_ARGUMENT;
    DeviceGuard g(plan->table->device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(plan->table->device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *a = nullptr, *k = nullptr, *r = nullptr;
    PFHE_TRY(st.upload(crt_glwe, len_glwe * sizeof(u32), &a));
    PFHE_TRY(st.upload(dcrt_ggsw, len_ggsw * sizeof(u32), &k));
    PFHE_TRY(st.alloc(len_result * sizeof(u32), &r));
    PFHE_TRY(pfhe_extprod32_mul_dcrt_ggsw_to_dev(plan, (const uint32_t *)a, len_glwe, (const uint32_t *)k, len_ggsw,
                                                 (uint32_t *)r, len_result, into_coeff_form, st.stream()));
    PFHE_TRY(st.download(result, r, len_result * sizeof(u32)));
    return st.finish();
    PFHE_GUARD_END
}

int pfhe_extprod32_add_dcrt_glev_mul_crt_poly_assign_dev(pfhe_extprod32_plan *plan, uint32_t *acc_dev, size_t len_acc,
                                                         const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                         const uint32_t *crt_poly_dev, size_t len_poly, void *stream) {
    PFHE_GUARD_BEGIN
    return glev32_common(plan, acc_dev, len_acc, dcrt_glev_dev, len_glev, crt_poly_dev, len_poly, true, false, stream);
    PFHE_GUARD_END
}

int pfhe_extprod32_glev_mul_crt_poly_to_dev(pfhe_extprod32_plan *plan, const uint32_t *dcrt_glev_dev, size_t len_glev,
                                            const uint32_t *crt_poly_dev, size_t len_poly, uint32_t *result_dev,
                                            size_t len_result, void *stream) {
    PFHE_GUARD_BEGIN
    return glev32_common(plan, result_dev, len_result, dcrt_glev_dev, len_glev, crt_poly_dev, len_poly, false, false, stream);
    PFHE_GUARD_END
}

int pfhe_extprod32_add_dcrt_glev_mul_big_uint_poly_assign_dev(pfhe_extprod32_plan *plan, uint32_t *acc_dev, size_t len_acc,
                                                              const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                              const uint32_t *big_uint_poly_dev, size_t len_poly,
                                                              void *stream) {
    PFHE_GUARD_BEGIN
    return glev32_common(plan, acc_dev, len_acc, dcrt_glev_dev, len_glev, big_uint_poly_dev, len_poly, true, true, stream);
    PFHE_GUARD_END
}

int pfhe_extprod32_glev_mul_big_uint_poly_to_dev(pfhe_extprod32_plan *plan, const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                 const uint32_t *big_uint_poly_dev, size_t len_poly, uint32_t *result_dev,
                                                 size_t len_result, void *stream) {
    PFHE_GUARD_BEGIN
    return glev32_common(plan, result_dev, len_result, dcrt_glev_dev, len_glev, big_uint_poly_dev, len_poly, false, true,
                         stream);
    PFHE_GUARD_END
}

}  // extern "C"
