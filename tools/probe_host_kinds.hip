// probe_host_kinds.hip — what tells memory the DRIVER allocated pinned (hipHostMalloc) from memory that is merely REGISTERED
// (hipHostRegister: a userptr mapping) on this runtime?  Prints every candidate signal for both, and for a torch-style
// sub-range.   hipcc --offload-arch=gfx950 -O2 -o probe_host_kinds probe_host_kinds.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

static void report(const char *what, void *p) {
    hipPointerAttribute_t at{};
    hipError_t e = hipPointerGetAttributes(&at, p);
    printf("%-28s %p: attributes rc=%d type=%d device=%d hostPointer=%p devicePointer=%p isManaged=%d allocationFlags=0x%x\n", what, p, (int)e,
           (int)at.type, at.device, at.hostPointer, at.devicePointer, at.isManaged, at.allocationFlags);
    (void)hipGetLastError();
    unsigned flags = 0;
    e = hipHostGetFlags(&flags, p);
    printf("    hipHostGetFlags rc=%d flags=0x%x\n", (int)e, flags);
    (void)hipGetLastError();
    void *d = nullptr;
    e = hipHostGetDevicePointer(&d, p, 0);
    printf("    hipHostGetDevicePointer rc=%d dev=%p (%s host pointer)\n", (int)e, d, d == p ? "==" : "!=");
    (void)hipGetLastError();
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    e = hipMemGetAddressRange(&base, &size, p);
    printf("    hipMemGetAddressRange rc=%d base=%p size=%zu\n", (int)e, base, size);
    (void)hipGetLastError();
    hipDeviceptr_t rb = nullptr;
    size_t rs = 0;
    hipError_t e1 = hipPointerGetAttribute(&rb, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, p);
    hipError_t e2 = hipPointerGetAttribute(&rs, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, p);
    printf("    RANGE_START_ADDR rc=%d %p  RANGE_SIZE rc=%d %zu\n", (int)e1, rb, (int)e2, rs);
    (void)hipGetLastError();
    unsigned mt = 0;
    e = hipPointerGetAttribute(&mt, HIP_POINTER_ATTRIBUTE_MEMORY_TYPE, p);
    printf("    MEMORY_TYPE rc=%d %u\n", (int)e, mt);
    (void)hipGetLastError();
    int mapped = -1;
    e = hipPointerGetAttribute(&mapped, HIP_POINTER_ATTRIBUTE_MAPPED, p);
    printf("    MAPPED rc=%d %d\n", (int)e, mapped);
    (void)hipGetLastError();
}

int main() {
    void *a = nullptr;
    hipHostMalloc(&a, 1 << 20, hipHostMallocDefault);
    report("hipHostMalloc", a);
    report("hipHostMalloc + 4096", (char *)a + 4096);
    void *m = nullptr;
    hipHostMalloc(&m, 1 << 20, hipHostMallocMapped | hipHostMallocPortable);
    report("hipHostMalloc mapped|portable", m);
    char *b = (char *)malloc(1 << 20);
    hipHostRegister(b, 1 << 20, hipHostRegisterDefault);
    report("hipHostRegister", b);
    report("hipHostRegister + 4096", b + 4096);
    hipHostUnregister(b);
    char *c = (char *)aligned_alloc(4096, 1 << 20);
    hipHostRegister(c, 1 << 20, hipHostRegisterMapped | hipHostRegisterPortable);
    report("hipHostRegister mapped", c);
    hipHostUnregister(c);
    report("malloc (pageable)", c);
    void *d = nullptr;
    hipMalloc(&d, 1 << 20);
    report("hipMalloc", d);
    return 0;
}
