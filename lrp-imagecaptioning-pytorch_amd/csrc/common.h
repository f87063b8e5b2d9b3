// Shared host-side helpers of liblrpx: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lrpx.h"

namespace lrpx {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LRPX_ELAUNCH;
    }
    return LRPX_OK;
}

#define LRPX_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            lrpx::set_error(__VA_ARGS__); \
            return LRPX_EINVAL;          \
        }                                \
    } while (0)

#define LRPX_TRY(expr)               \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != LRPX_OK) return rc__; \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace lrpx
