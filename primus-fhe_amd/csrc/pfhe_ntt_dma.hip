// pfhe_ntt_dma.hip — LDS-DMA form of the forward pipelined kernel (round 5 experiment, selected by PFHE_PIPE_DMA).
//
// The same (block of tile k-1, chunk of tile k) pairs as ntt_pipe_fwd_kernel (pfhe_ntt.hip), but a workgroup of 512
// threads (8 coefficients each, BlockCfg<12, 3>) walks over `per_wg` pairs and owns TWO LDS images: while pair i computes
// in one image, block i+1 arrives in the other by global_load_lds_dwordx4 — the gfx950 load that writes LDS directly,
// 1 KiB per wave instruction, no destination registers and no ds_write — so that no workgroup waits for its own block with
// nothing else to do (the plain pipelined kernel's first instruction is that wait: 14 % of a workgroup's life,
// profiles/r02_pipe_kernel_phase_stamps.txt).  Two workgroups x 72 KiB per CU = the same sixteen waves.
//   image layout: the DMA lands a block in NATURAL order (wave-uniform LDS base + lane x 16 bytes is all the instruction
//   can address, so no padding); the first register layout <9> reads it as img[(k << 9) + lt], consecutive lanes on
//   consecutive words: conflict-free without a swizzle.  After a barrier the same 36 KiB serve the block's exchanges and
//   its output staging in the padded layout of the other block kernels.
//   ordering: a wave's DMA pieces are older than every other vector-memory operation of the iteration, and the wave drains
//   its counter (s_waitcnt vmcnt(0), free at that point: the pieces were issued twelve stages earlier) in front of the
//   block's stores; the barrier at the top of the next iteration then orders every wave's pieces before the first ds_read
//   of the image (LDS-DMA data is visible to other waves only after the issuer's vmcnt wait and a barrier), and every
//   wave's last read of the other image (its output staging one iteration earlier) before the DMA that overwrites it.
// The DMA is inline asm: the compiler neither counts it nor drains it at its own barriers (cdna_hip_programming.md §5,
// "Pipelining across barriers").  M0 (the LDS base of the DMA) is saved and restored inside the statement.
// Reference work being computed: scalar_forward_transform, primus_ntt/src/ntt/prime64/scalar/transform.rs:13-141.
#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"

namespace pfhe {

__device__ __forceinline__ void glds16_nt(const void *gsrc, u32 lds_dst) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// experiment modes (bits 16.. of the launcher's per_wg argument)
constexpr u32 kDmaInterleaved = 1u << 16;  // pair i of workgroup g is g + i * gridDim (default: g * per_wg + i, consecutive)
constexpr u32 kDmaChunkAll = 1u << 17;     // the strided chunk as 512 columns by all eight waves in even iterations
                                           // (default: 256 columns by waves 0-3 / 4-7 in even / odd iterations)
constexpr u32 kDmaOneImage = 1u << 18;     // the 256-thread / 16-coefficient / one-image form (LOGE = 4)
// bits 20..27: start-up stagger — workgroup g sleeps (g * 40503 >> 4 & 255) * stagger * 64 cycles before its first pair

// LOGE = 3: 512 threads, two images, the DMA of block i + 1 issued at the top of iteration i (lands during the block's stages).
// LOGE = 4: 256 threads of 16 coefficients (the plain pipelined kernel's core and its four workgroups per CU), ONE image:
//           the DMA of block i + 1 is issued once every wave has read block i's results out of the image (output staging)
//           and lands during the strided chunk's four stages.
template <class A, int LOGE>
__global__ __launch_bounds__(1 << (12 - LOGE)) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_pipe_fwd_dma_kernel(
    u64 *__restrict__ blk_data, u64 blk_total, u64 *__restrict__ str_data, u64 str_total,
    const NttPrime *__restrict__ primes, u32 L, u32 lazy, u32 mode) {
    constexpr int LOGB = 12, K = 4;
    using Cfg = BlockCfg<LOGB, LOGE>;
    constexpr u32 log_n = LOGB + K, n = 1u << log_n;
    constexpr int NV = Cfg::E / 2;
    constexpr bool kTwoImages = LOGE == 3;
    constexpr u32 kWaves = Cfg::TPB / 64, kPieces = 32 / kWaves;  // 1 KiB pieces of a block per wave
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];  // one or two images of Cfg::LDS_WORDS
    const u32 tid = threadIdx.x;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) u64 *)lds_raw;
    const u64 total = blk_total > str_total ? blk_total : str_total;
    const bool inter = (mode & kDmaInterleaved) != 0, chunk_all = kTwoImages && (mode & kDmaChunkAll) != 0;
    // the launch's pairs are dealt evenly over the grid (sizes differ by at most one): a workgroup more than the CUs hold
    // at once, or one with a longer run than the others, would be a tail as long as a whole run
    const u64 p0 = inter ? (u64)blockIdx.x : (u64)blockIdx.x * total / gridDim.x;
    const u64 step = inter ? (u64)gridDim.x : 1ull;
    const u32 per_wg = inter ? (u32)((total - blockIdx.x + gridDim.x - 1) / gridDim.x)
                             : (u32)(((u64)blockIdx.x + 1) * total / gridDim.x - p0);
    if (p0 >= total || per_wg == 0) return;
    if (const u32 stagger = (mode >> 20) & 255u) {
        const u32 units = ((blockIdx.x * 40503u) >> 4) & 255u;
        for (u32 t = 0; t < units * stagger; ++t) __builtin_amdgcn_s_sleep(1);
    }
    const auto dma_block = [&](u64 p, u32 image) {
        const char *g = (const char *)(blk_data + (p << LOGB)) + (wave * kPieces) * 1024u + (tid & 63u) * 16u;
        const u32 dst = lds0 + image * (u32)(Cfg::LDS_WORDS * sizeof(u64)) + (wave * kPieces) * 1024u;
#pragma unroll
        for (u32 j = 0; j < kPieces; ++j) glds16_nt(g + j * 1024u, __builtin_amdgcn_readfirstlane(dst + j * 1024u));
    };
    if (p0 < blk_total) {
        dma_block(p0, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (u32 i = 0; i < per_wg; ++i) {
        const u64 p = p0 + i * step;
        const bool has_blk = p < blk_total;
        const bool more = i + 1 < per_wg && p + step < blk_total;
        // the strided chunk this thread takes part in during this iteration: chunk id ps, column t of its 256
        bool mine;
        u64 ps;
        if (chunk_all) {  // pairs 2j and 2j + 1 of this workgroup: both chunks in iteration 2j, one per half of the threads
            const u32 half = wave >> 2;  // (wave-uniform, like everything derived from it: the prime's constants stay scalar)
            ps = p0 + ((i & ~1u) + half) * step;
            mine = (i & 1u) == 0u && (i + half) < per_wg && ps < str_total;
        } else if (kTwoImages) {  // waves 0-3 in even iterations, waves 4-7 in odd ones
            ps = p;
            mine = p < str_total && (((wave >> 2) ^ i) & 1u) == 0u;
        } else {
            ps = p;
            mine = p < str_total;
        }
        const u32 col = tid & 255u;
        const u32 cur = kTwoImages ? (i & 1u) : 0u;
        u64 *__restrict__ img = lds_raw + (size_t)cur * Cfg::LDS_WORDS;
        // every wave's pieces of block i have landed (its vmcnt wait of the previous iteration / the prologue), and every
        // wave has finished reading the image the next DMA overwrites
        __builtin_amdgcn_s_barrier();
        u64 x[Cfg::E];
        {
            const u32 lt = opaque_tid();
#pragma unroll
            for (int k = 0; k < Cfg::E; ++k) x[k] = has_blk ? img[((u32)k << (LOGB - LOGE)) + lt] : 0ull;
        }
        if constexpr (kTwoImages) {
            if (more) dma_block(p + step, cur ^ 1u);
        }
        // the strided chunk's loads: in flight during the block's twelve stages
        u64 *__restrict__ sp = str_data + (ps >> 4) * n + (ps & 15) * 256u + col;
        u64 sx[1 << K][1];
        if (mine) {
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) sx[k][0] = __builtin_nontemporal_load(sp + ((u64)k << LOGB));
        }
        if (has_blk) {
            const A ar(primes + (p >> 4) % L);
            const u32 eblk = (u32)(p & 15) << LOGB;
            block_forward_core<A, LOGB, true, LOGE>(ar, x, img, n, eblk, opaque_tid(), lazy != 0);
            const u32 lt = opaque_tid();
            lds_put_layout<0, LOGE>(x, img, lt);
            __syncthreads();
            u64x2 io[NV];
            lds_get_vectors<LOGB, LOGE>(io, img, lt);
            // two images: this wave's DMA pieces of block i + 1 (and the chunk's loads) are complete before anything younger
            // is issued
            if constexpr (kTwoImages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_block_vectors<LOGB, LOGE, true>(io, blk_data + (p << LOGB), lt);
        } else if constexpr (kTwoImages) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!kTwoImages) {
            if (more) {  // one image: every wave has its results in registers (the stores above read them) before the DMA
                __builtin_amdgcn_s_barrier();
                dma_block(p + step, 0u);
            }
        }
        if (mine) {
            const A ars(primes + (ps >> 4) % L);
            strided_forward_regs<A, K, 1, true>(ars, sx, n, 0u, LOGB);
            // one image: the DMA issued above has had the four stages to land; its completion is waited for here, in front
            // of this wave's last stores, so that the next iteration opens with the barrier alone
            if constexpr (!kTwoImages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) gstore<kPipeIntermediateNt>(sp + ((u64)k << LOGB), sx[k][0]);
        } else if constexpr (!kTwoImages) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

int launch_pipe_fwd_dma(int arith, u64 *blk_data, u64 blk_total, u64 *str_data, u64 str_total, const NttPrime *primes, u32 L,
                        bool lazy, int per_wg_mode, hipStream_t s) {
    // low 16 bits: workgroups of the launch in units of the device's resident capacity (two per CU); they share the
    // launch's pairs evenly
    const u32 gens = (u32)per_wg_mode & 0xffffu, mode = (u32)per_wg_mode & ~0xffffu;
    if (gens == 0) return PFHE_ERR_BAD_ARGUMENT;
    const u64 total = blk_total > str_total ? blk_total : str_total;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    const bool one_image = (mode & kDmaOneImage) != 0;
    u64 grid = (u64)cus * (one_image ? 4 : 2) * gens;
    if (grid > total) grid = total;
    if (grid == 0) return PFHE_OK;
    if (grid > 0x7fffffffull) return PFHE_ERR_BAD_LENGTH;
    constexpr size_t lds2 = 2 * sizeof(u64) * (size_t)(BlockCfg<12, 3>::LDS_WORDS);
    constexpr size_t lds1 = sizeof(u64) * (size_t)(BlockCfg<12, 4>::LDS_WORDS);
    const u32 lz = lazy ? 1u : 0u;
    if (arith == kArithPm && one_image)
        hipLaunchKernelGGL((ntt_pipe_fwd_dma_kernel<PmArith, 4>), dim3((u32)grid), dim3(256), lds1, s, blk_data, blk_total,
                           str_data, str_total, primes, L, lz, mode);
    else if (arith == kArithMont && one_image)
        hipLaunchKernelGGL((ntt_pipe_fwd_dma_kernel<MontArith, 4>), dim3((u32)grid), dim3(256), lds1, s, blk_data, blk_total,
                           str_data, str_total, primes, L, lz, mode);
    else if (arith == kArithPm)
        hipLaunchKernelGGL((ntt_pipe_fwd_dma_kernel<PmArith, 3>), dim3((u32)grid), dim3(512), lds2, s, blk_data, blk_total,
                           str_data, str_total, primes, L, lz, mode);
    else if (arith == kArithMont)
        hipLaunchKernelGGL((ntt_pipe_fwd_dma_kernel<MontArith, 3>), dim3((u32)grid), dim3(512), lds2, s, blk_data, blk_total,
                           str_data, str_total, primes, L, lz, mode);
    else
        return PFHE_ERR_UNSUPPORTED;
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe
