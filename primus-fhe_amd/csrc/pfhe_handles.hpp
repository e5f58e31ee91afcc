// pfhe_handles.hpp — what a table handle owns.
#pragma once
#include <memory>
#include <vector>

#include "pfhe_common.hpp"
#include "pfhe_ntt_device.hpp"

namespace pfhe {

// One or more per-prime NTT tables living on one GPU (U64NttTable = 1, U64DcrtTable = L).
struct TableSet {
    int device = 0;
    u32 log_n = 0;
    size_t n = 1;
    u32 L = 0;
    bool pm = false;                       // every prime has the pseudo-Mersenne shape
    int ntt_arith = 0;                     // policy of the plain transforms: kArithPm, kArithMont (generic primes below 2^61) or kArithShoup
    // tuning switches, read from the environment once, when the handle is created
    NttTuning tune;
    std::vector<NttPrime> primes;          // host copies (device pointers inside)
    const NttPrime *primes_dev = nullptr;  // the same array on the device
    const u64 *moduli_dev = nullptr;
    std::vector<u64> roots, inv_roots;
    std::vector<void *> allocations;
    ~TableSet();
};

int make_table_set(u32 log_n, const u64 *moduli, size_t count, int device, std::unique_ptr<TableSet> &out);
int transform_dev(const TableSet &t, u64 *data, size_t len, bool inverse, bool lazy, hipStream_t s);
int transform_host(const TableSet &t, u64 *host, size_t len, bool inverse, bool lazy);
size_t stage_bounce_max();   // PFHE_STAGE_BOUNCE_MAX: largest slice that goes through the pinned bounce buffer
bool stage_zero_copy();      // PFHE_STAGE_ZERO_COPY=0 clears it: one-piece host slices are copied instead of mapped
size_t stage_chunk_bytes();  // PFHE_STAGE_CHUNK: bytes per piece of a pipelined host-pointer transform
int pointwise(const TableSet &t, int mode, u64 *acc, const u64 *a, size_t len_a, const u64 *b, size_t len_b,
              hipStream_t s);

}  // namespace pfhe
