// Shared host-side helpers of liblrpx: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <initializer_list>
#include <mutex>

#include "../../include/lrpx.h"

#include "build_guard.h"

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

// ---- caller pointers ------------------------------------------------------------------------------------------
// The C ABI takes raw device pointers; a pageable host pointer (a CPU tensor's data_ptr) handed to a kernel is a GPU page fault, not
// an error code.  check_dev_ptrs asks the runtime what each non-null pointer is (hipPointerGetAttributes, ~1 us each): device /
// managed memory of the CURRENT device or pinned host memory pass; anything else returns LRPX_EINVAL with the entry point and the
// argument's name in lrpx_last_error_string().  The heavy entry points (milliseconds of work: the VGG16 chains, the packers, the
// layout converters, the rule classes' kernels) validate on every call; the microsecond-scale decoder steps only when
// LRPX_CHECK_PTRS=1 is in the environment (latched at first use).
struct PtrArg { const void* p; const char* name; };
int check_dev_ptrs(const char* fn, std::initializer_list<PtrArg> args);     // (lrpx_core.hip)
bool ptr_checks_all();
#define LRPX_CHECK_PTRS(fn, ...) LRPX_TRY(lrpx::check_dev_ptrs(fn, {__VA_ARGS__}))
#define LRPX_CHECK_PTRS_OPT(fn, ...)                                                   \
    do {                                                                               \
        if (lrpx::ptr_checks_all()) LRPX_TRY(lrpx::check_dev_ptrs(fn, {__VA_ARGS__})); \
    } while (0)

// elementwise producers of S that also record max|S| per map for an f16x3 consumer (lrpx_core.hip)
int maxpool_relevance_amax(const float* x, const float* r_out, const float* zdiv, const int32_t* map2img, float* r_in,
                           float* s_out, int n_maps, int h_out, int w_out, int c, int s_chunk, unsigned* amax,
                           hipStream_t stream);
int divide_stab_amax(const float* r, const float* z, const int32_t* map2img, float* s, int n_maps, long pix_c, int stab,
                     unsigned* amax, hipStream_t stream);
int pool_winner_blk(const float* x, const float* z, float* xzw, uint8_t* am, float* xzw_blk, int n, int h_out, int w_out, int c, hipStream_t stream,
                    float* y_pool = nullptr);     // y_pool: also the pooled activations (the forward pass)
int divide_safe_blk(const float* r, const float* z, float* s, float* s_blk, int n_img, int pix, int c, hipStream_t stream);
int divide_stab_blocked(const float* r, const float* z, const int32_t* map2img, float* s_blk, int n_maps, int pix, int c,
                        unsigned* amax, hipStream_t stream);

// max|x| over a flat tensor, atomically max-ed into *out as float bits (lrpx_core.hip)
int amax_flat(const float* x, long n, unsigned* out, hipStream_t stream);

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel instantiation.  HIP function attributes are PER DEVICE, and one
// host process may drive several GPUs from several threads (include/lrpx.h THREADING), so the attribute is set exactly once
// per (instantiation, device ordinal) even when several host threads make their first launch of it at the same time
// (no globals except an init-once cache behind a mutex).  `LdsOnce` is the caller's function-local static
// (one per template instantiation).
struct LdsOnce {
    static constexpr int MAX_DEV = 64;
    std::once_flag once[MAX_DEV];
    hipError_t res[MAX_DEV] = {};
};
template <typename K>
inline int reserve_lds_once(LdsOnce& st, K kern, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LdsOnce::MAX_DEV) {
        set_error("%s: cannot identify the current device", what);
        return LRPX_ELAUNCH;
    }
    std::call_once(st.once[dev], [&] {
        st.res[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    });
    if (st.res[dev] != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS on device %d", what, bytes, dev);
        return LRPX_ELAUNCH;
    }
    return LRPX_OK;
}

// A/B switches of the shipped dispatch paths: ONE getenv pass per process, latched on first use, so that paired decisions
// (LRPX_FIRST_VALU picks both the first-layer kernel and the chunk width its producer writes) cannot diverge.  Defaults are
// what every number in DESIGN.md is measured with; the non-default values are exercised by tests/test_gpu_switches.py.
struct Switches {
    int wide;            // LRPX_WIDE (bit mask, default 7): 8-wave relevance kernels for 56/28 (1), 14 (2), pooled-input 56/28 (4)
    int fwd_ksplit14;    // LRPX_FWD_KSPLIT (default 8): K ranges per tile of the 14x14 forward layers (1 = unsplit)
    int fwd_ksplit28;    // LRPX_FWD_KSPLIT28 (default 1 = unsplit; 2 / 4 built and tested): ... of the 28x28 forward layers
    int fwd_wide;        // LRPX_FWD_WIDE (bit mask, default 0): 8-wave forward kernels for 112 (1), 56 (2), 28 (4), K-split 14 (8)
    int conv11_f16;      // LRPX_CONV11_F16 (default 1): conv1_1 of the forward trace on the f16x3 kernel; 0 = fp32 MFMA kernel
    int first_valu;      // LRPX_FIRST_VALU: first-layer rule on the VALU kernel (and S from conv1_2 in 16-channel chunks)
    int s21_nhwc;        // LRPX_S21_NHWC: S between conv2_2 and conv2_1 as NHWC instead of 16-channel chunks
    int guided_poolbwd;  // LRPX_GUIDED_POOLBWD: image-gradient chain with pool-backward kernels
    int dense_wide;      // LRPX_DENSE_WIDE (default 0): 256-row 8-wave tiles for the many-row dense f16x3 GEMM (1), 128-row tiles (0)
    int dense_ks_rel;    // LRPX_DENSE_KS_REL: the few-row epsilon rule (the decoders' lock-steps in the exact modes) with K split over four waves (dense_ks_kernel)
    int dense_n256;      // LRPX_DENSE_N256 (default 1): 128 x 256 tiles with the waves side by side for the many-row dense f16x3 GEMM
    int dense_rt;        // LRPX_DENSE_RT (default 0 = by the grid): row tiles per wave of that kernel, 4 (128-row tiles) or 3 (96-row tiles)
    int dense_1wave;     // LRPX_DENSE_1WAVE: PLAIN few-row GEMMs without the 4-wave K split
    int linear_valu;     // LRPX_LINEAR_VALU: the VALU skinny linear instead of the fp32-MFMA one
    int b6_fwd_ksplit28; // LRPX_B6_FWD_KSPLIT28 (default 4): K ranges per tile of the 28x28 layers of the exact-split (conv mode 1) forward trace
    int b6_fwd_ksplit56; // LRPX_B6_FWD_KSPLIT56 (default 2): ... of the 56x56 layers
    int b6_wide;         // LRPX_B6_WIDE (bit mask): 8-wave conv-mode-1 relevance kernels for 56/28 (1), 14 (2), pooled-input 56/28 (4)
    int b6_rel_ksplit14; // LRPX_B6_REL_KSPLIT14 (default 2; 1 = unsplit, 4 built): K ranges per tile of the 14x14 RELEVANCE layers of conv mode 1 (partial sums + rel_mul_finish)
    int x6_legacy;       // LRPX_X6_LEGACY: conv mode 1 on round 1's flow (conv_bf16x6.h with EPI_REL + pool kernels) instead of the fused B6 kernels
};
const Switches& switches();      // (lrpx_core.hip)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace lrpx
