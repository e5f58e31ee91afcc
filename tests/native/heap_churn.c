/* heap_churn.c — test infrastructure (tests/test_gpu_host_hazard.py, tools/stress_host_slice.py): a host thread that keeps
 * the C heap in motion while another thread drives the library's host-pointer entry points.  Blocks of three size classes
 * come and go in random order: small ones (the thread's malloc arena grows and trims by mprotect / madvise), medium ones
 * around glibc's mmap threshold, large ones (always mmap + munmap: the same virtual addresses reappear with other pages
 * behind them).  Every block is written to (first and last page fully, one byte per page in between) and carries a guard
 * pattern that is checked before it is freed: returns the number of blocks whose pattern was damaged. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

enum { SLOTS = 48 };

uint64_t heap_churn(uint64_t seed, uint64_t rounds, volatile int *stop) {
    unsigned char *slot[SLOTS] = {0};
    size_t size[SLOTS] = {0};
    unsigned char tag[SLOTS] = {0};
    uint64_t damaged = 0;
    for (uint64_t r = 0; r < rounds && !(stop && *stop); ++r) {
        const uint64_t x = splitmix(&seed);
        const int i = (int)(x % SLOTS);
        if (slot[i]) {
            const size_t n = size[i];
            if (slot[i][0] != tag[i] || slot[i][n - 1] != tag[i] || slot[i][n / 2] != tag[i]) ++damaged;
            free(slot[i]);
            slot[i] = 0;
        }
        const unsigned cls = (unsigned)((x >> 8) & 7);
        size_t n;
        if (cls < 4) n = 64 + (size_t)((x >> 16) % (64u << 10));                 /* arena */
        else if (cls < 6) n = (64u << 10) + (size_t)((x >> 16) % (2u << 20));    /* around the mmap threshold */
        else n = (2u << 20) + (size_t)((x >> 16) % (22u << 20));                /* mmap / munmap */
        unsigned char *p = (unsigned char *)malloc(n);
        if (!p) continue;
        const unsigned char t = (unsigned char)(x >> 56) | 1;
        const size_t edge = n < 4096 ? n : 4096;
        memset(p, t, edge);
        memset(p + n - edge, t, edge);
        for (size_t o = 4096; o + 4096 < n; o += 4096) p[o] = t;
        p[n / 2] = t;
        slot[i] = p;
        size[i] = n;
        tag[i] = t;
    }
    for (int i = 0; i < SLOTS; ++i) {
        if (slot[i]) {
            const size_t n = size[i];
            if (slot[i][0] != tag[i] || slot[i][n - 1] != tag[i] || slot[i][n / 2] != tag[i]) ++damaged;
            free(slot[i]);
        }
    }
    return damaged;
}
