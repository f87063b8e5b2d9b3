// Shared host-side helpers of liblrpx: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

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

// elementwise producers of S that also record max|S| per map for an f16x3 consumer (lrpx_core.hip)
int maxpool_relevance_amax(const float* x, const float* r_out, const float* zdiv, const int32_t* map2img, float* r_in,
                           float* s_out, int n_maps, int h_out, int w_out, int c, int s_chunk, unsigned* amax,
                           hipStream_t stream);
int divide_stab_amax(const float* r, const float* z, const int32_t* map2img, float* s, int n_maps, long pix_c, int stab,
                     unsigned* amax, hipStream_t stream);

// max|x| over a flat tensor, atomically max-ed into *out as float bits (lrpx_core.hip)
int amax_flat(const float* x, long n, unsigned* out, hipStream_t stream);

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel instantiation, set exactly once even when several host threads
// make their first launch of it at the same time (no globals except an init-once cache behind a mutex).
// `once` / `res` are the caller's function-local statics (one pair per template instantiation).
template <typename K>
inline int reserve_lds_once(std::once_flag& once, hipError_t& res, K kern, int bytes, const char* what) {
    std::call_once(once, [&] {
        res = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    });
    if (res != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS", what, bytes);
        return LRPX_ELAUNCH;
    }
    return LRPX_OK;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace lrpx
