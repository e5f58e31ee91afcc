// microbench11_residency.hip — how many workgroups of a given LDS size does a CU of the MI355X hold at once?
// Round 4 found that five workgroups of 32 KiB "need all 160 KiB and the fifth does not become resident"; this measures it in
// isolation: every workgroup (256 threads, few registers) sleeps a fixed time; a grid of W workgroups per CU takes one
// sleep period if all W are resident at once and two if not.   hipcc --offload-arch=gfx950 -O2 -o microbench11 microbench11_residency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void sleeper(unsigned long long ticks, unsigned *out) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;  // the allocation is used
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 7) & 255];
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int max_lds = 0;
    hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    printf("CUs %d, max LDS per workgroup %d bytes\n", cus, max_lds);
    unsigned *out;
    hipMalloc(&out, sizeof(unsigned) * cus * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned long long ticks = 20000000ull / 100;  // s_memtime / cycle counter at 100 MHz: 2 ms
    hipFuncSetAttribute((const void *)sleeper, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {65536, 40960, 36864, 32768, 32256, 31744, 30720, 28672, 24576, 20480, 18432, 16384};
    for (int lds : sizes) {
        for (int per_cu : {1, 2, 3, 4, 5, 6, 8, 9, 10}) {
            if ((long)per_cu * lds > 200 * 1024) continue;
            hipLaunchKernelGGL(sleeper, dim3(cus * per_cu), dim3(256), lds, 0, ticks, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(sleeper, dim3(cus * per_cu), dim3(256), lds, 0, ticks, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("LDS %6d B x %2d workgroups per CU (%7d B): %6.2f ms  -> %s\n", lds, per_cu, lds * per_cu, ms,
                   ms < 3.0f ? "all resident" : "NOT all resident");
        }
    }
    return 0;
}
