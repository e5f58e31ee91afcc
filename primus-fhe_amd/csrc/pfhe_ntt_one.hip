// pfhe_ntt_one.hip — round-5 experiment (PFHE_PIPE_ONE): the forward pipelined transform of N = 2^16 as ONE launch.
//
// The tree runs tiles + 1 launches of ntt_pipe_fwd_kernel: the kernel boundary is what orders "strided pass of a
// polynomial" before "block pass of the same polynomial".  Here workgroup i runs the strided chunk of pair i and the block of
// pair i - lag in one launch of T + lag workgroups; what the kernel boundary gave has to be supplied by hand:
//   * ordering: the dispatcher starts workgroups in index order, so a block's sixteen chunks were STARTED `lag` workgroups
//     earlier — in practice long finished, by the programming model not guaranteed.  FLAGS adds the guarantee: every chunk
//     workgroup adds one to a counter of its polynomial once its stores are acknowledged, every block workgroup reads its
//     polynomial's counter together with its data and retries while it is short of sixteen (never observed to spin when
//     lag >= 2048).  A producer never waits for anything, so waiting consumers cannot deadlock.
//   * visibility: an XCD's L2 is not coherent with another's inside one launch.  COH writes the intermediate with
//     agent-scope (write-through, sc1) stores and reads it with agent-scope (L2-bypassing) loads; the transform's input and
//     output keep their non-temporal accesses.
// Work being computed: scalar_forward_transform, primus_ntt/src/ntt/prime64/scalar/transform.rs:13-141.
#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"

namespace pfhe {

template <bool COH>
__device__ __forceinline__ u64 load_mid(const u64 *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __builtin_nontemporal_load(p);
}
template <bool COH>
__device__ __forceinline__ void store_mid(u64 *p, u64 v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// flags[poly]: chunks of limb-polynomial `poly` whose strided pass is in memory (16 = all); zeroed by the launcher
template <class A, bool COH, bool FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_pipe_one_kernel(
    u64 *__restrict__ data, u64 total, u64 lag, const NttPrime *__restrict__ primes, u32 L, u32 lazy, u32 *__restrict__ flags) {
    constexpr int LOGB = 12, K = 4, LOGE = 4;
    using Cfg = BlockCfg<LOGB, LOGE>;
    constexpr u32 log_n = LOGB + K, n = 1u << log_n;
    constexpr int NV = Cfg::E / 2;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 i = blockIdx.x;
    const bool has_str = i < total, has_blk = i >= lag;  // (grid = total + lag)
    const u64 pb = i - lag;                               // block pair
    const u32 lt = threadIdx.x;
    u64 x[Cfg::E];
    u64 *__restrict__ gb = data + (pb << LOGB);
    if (has_blk) {
        if constexpr (FLAGS) {
            // the polynomial's counter, read (L2-bypassing) together with the data; short of 16 -> poll, then read again
            const u32 *f = flags + (pb >> 4);
            u32 seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < Cfg::E; ++k) x[k] = load_mid<COH>(gb + ((u32)k << (LOGB - LOGE)) + lt);
            seen = __builtin_amdgcn_readfirstlane(seen);
            if (seen < 16u) {
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) __builtin_amdgcn_s_sleep(8);
#pragma unroll
                for (int k = 0; k < Cfg::E; ++k) x[k] = load_mid<COH>(gb + ((u32)k << (LOGB - LOGE)) + lt);
            }
        } else {
#pragma unroll
            for (int k = 0; k < Cfg::E; ++k) x[k] = load_mid<COH>(gb + ((u32)k << (LOGB - LOGE)) + lt);
        }
    }
    u64 *__restrict__ sp = data + (i >> 4) * n + (i & 15) * 256u + lt;
    u64 sx[1 << K][1];
    if (has_str) {
#pragma unroll
        for (int k = 0; k < (1 << K); ++k) sx[k][0] = __builtin_nontemporal_load(sp + ((u64)k << LOGB));
    }
    if (has_blk) {
        const A ar(primes + (pb >> 4) % L);
        block_forward_core<A, LOGB, false, LOGE>(ar, x, lds, n, (u32)(pb & 15) << LOGB, lt, lazy != 0);
        lds_put_layout<0, LOGE>(x, lds, lt);
        __syncthreads();
        u64x2 io[NV];
        lds_get_vectors<LOGB, LOGE>(io, lds, lt);
        store_block_vectors<LOGB, LOGE, true>(io, gb, lt);
    }
    if (has_str) {
        const A ar(primes + (i >> 4) % L);
        strided_forward_regs<A, K, 1, true>(ar, sx, n, 0u, LOGB);
#pragma unroll
        for (int k = 0; k < (1 << K); ++k) store_mid<COH>(sp + ((u64)k << LOGB), sx[k][0]);
        if constexpr (FLAGS) {
            // every wave's stores acknowledged (write-through: in memory), then ONE add for the workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + (i >> 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// mode: bit 0 COH, bit 1 FLAGS.  `flags` must hold total / 16 zeroed u32 when FLAGS.
int launch_pipe_one(int arith, u64 *data, u64 npolys, const NttPrime *primes, u32 L, bool lazy, u64 lag, int mode, u32 *flags,
                    hipStream_t s) {
    if (arith != kArithPm) return PFHE_ERR_UNSUPPORTED;
    const u64 total = npolys << 4;
    if (lag < 16) lag = 16;
    lag = (lag + 15) & ~15ull;  // whole polynomials
    const u64 grid = total + lag;
    if (grid > 0x7fffffffull) return PFHE_ERR_BAD_LENGTH;
    constexpr size_t lds = sizeof(u64) * (size_t)(BlockCfg<12, 4>::LDS_WORDS);
    const u32 lz = lazy ? 1u : 0u;
    const dim3 g((u32)grid), t(256);
    switch (mode & 3) {
        case 0: hipLaunchKernelGGL((ntt_pipe_one_kernel<PmArith, false, false>), g, t, lds, s, data, total, lag, primes, L, lz, flags); break;
        case 1: hipLaunchKernelGGL((ntt_pipe_one_kernel<PmArith, true, false>), g, t, lds, s, data, total, lag, primes, L, lz, flags); break;
        case 2: hipLaunchKernelGGL((ntt_pipe_one_kernel<PmArith, false, true>), g, t, lds, s, data, total, lag, primes, L, lz, flags); break;
        default: hipLaunchKernelGGL((ntt_pipe_one_kernel<PmArith, true, true>), g, t, lds, s, data, total, lag, primes, L, lz, flags); break;
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe
