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
    std::printf("host sanitize run ok\n");
    return 0;
}
