// pfhe_capi_internal.hpp — helpers shared by the extern "C" translation units.
#pragma once
#include <new>

#include "pfhe_common.hpp"
#include "pfhe_handles.hpp"

#define PFHE_GUARD_BEGIN try {
#define PFHE_GUARD_END                                   \
    }                                                    \
    catch (const std::bad_alloc &) {                     \
        ::pfhe::set_last_error("out of host memory");    \
        return PFHE_ERR_HIP;                             \
    }                                                    \
    catch (...) {                                        \
        ::pfhe::set_last_error("unexpected C++ exception"); \
        return PFHE_ERR_HIP;                             \
    }

struct pfhe_dcrt;
struct pfhe_dcrt32;
namespace pfhe {
int capi_check_device(int device);
const TableSet *capi_table_of(const pfhe_dcrt *t);
const TableSet *capi_table32_of(const pfhe_dcrt32 *t);
}  // namespace pfhe
