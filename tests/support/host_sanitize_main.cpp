// Host-side product code (table / RNS / gadget constant construction) under ASan + UBSan.
#include <cstdio>
#include <string>
#include "pfhe_common.hpp"
#include "pfhe_rns.hpp"
namespace pfhe {
void set_last_error(const std::string &) {}
int hip_fail(hipError_t, const char *, const char *, int) { return PFHE_ERR_HIP; }
}
using namespace pfhe;
int main() {
    const u64 Q61[3] = {2305843009211596801ull, 2305843009210023937ull, 2305843009208713217ull};
    for (u32 log_n = 0; log_n <= 14; ++log_n) {
        HostTable t;
        if (build_host_table(log_n, Q61[log_n % 3], t) != PFHE_OK) return 1;
        if (t.fwd.size() != ((size_t)1 << log_n)) return 2;
    }
    HostTable bad;
    if (build_host_table(20, 97, bad) != PFHE_ERR_NO_PRIMITIVE_ROOT) return 3;
    RnsHost r;
    if (build_rns(Q61, 3, r) != PFHE_OK) return 4;
    const u64 two[2] = {21, 35};
    RnsHost r2;
    if (build_rns(two, 2, r2) != PFHE_ERR_COPRIME) return 5;
    for (u32 lb : {1u, 7u, 13u, 30u, 31u, 45u}) {
        BasisHost b;
        if (build_basis(r, lb, 0, b) != PFHE_OK) return 6;
        BasisHost b2;
        if (build_basis(r, lb, 1, b2) != PFHE_OK) return 7;
    }
    // round 6: bases wider than the by-value form (9 and 32 moduli: host constants only, the device table is uploaded by
    // the caller), and the <u32> instantiation (word_bits = 32: moduli below 2^30, 32-bit limb counts)
    {
        u64 wide[32];
        u64 q = (1ull << 30) - 1;
        u32 found = 0;
        auto coprime_to_all = [&](u64 c) {
            for (u32 i = 0; i < found; ++i) {
                u64 a = wide[i], b = c;
                while (b) { u64 t = a % b; a = b; b = t; }
                if (a != 1) return false;
            }
            return true;
        };
        for (; found < 32; q -= 2) if (coprime_to_all(q)) wide[found++] = q;
        for (size_t count : {9u, 16u, 32u}) {
            for (u32 bits : {64u, 32u}) {
                RnsHost w;
                if (build_rns(wide, count, w, bits) != PFHE_OK) return 8;
                if (w.moduli.size() != count || w.punct.size() != count * w.par.dev.value_len) return 9;
                for (u32 lb : {1u, 15u, 31u}) {
                    BasisHost b;
                    if (build_basis(w, lb, 0, b) != PFHE_OK) return 10;
                }
            }
        }
        RnsHost too_many;
        u64 wide33[33];
        for (u32 i = 0; i < 32; ++i) wide33[i] = wide[i];
        wide33[32] = 2;   // coprime to every odd modulus above
        if (build_rns(wide33, 33, too_many) != PFHE_ERR_UNSUPPORTED) return 11;
        RnsHost big32;
        const u64 over[1] = {1ull << 30};
        if (build_rns(over, 1, big32, 32) != PFHE_ERR_UNREPRESENTABLE_MODULUS) return 12;
        BasisHost b32;
        RnsHost r32;
        if (build_rns(wide, 3, r32, 32) != PFHE_OK) return 13;
        if (build_basis(r32, 32, 0, b32) == PFHE_OK) return 14;   // log_basis below the word width (basis.rs:51)
    }
    std::printf("host sanitize run ok\n");
    return 0;
}
