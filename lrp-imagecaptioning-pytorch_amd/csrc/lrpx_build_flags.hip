// lrpx_build_flags(): which timing-experiment / profiling switches this library was compiled with ("" = release build).
// Kept in a file of its own (no device code, no other headers) so that tests/test_abi.py can compile it alone with the
// host compiler and check both directions of build_guard.h.
#include "build_guard.h"
#include "../../include/lrpx.h"

extern "C" {

#define LRPX_STR2(x) #x
#define LRPX_STR(x) LRPX_STR2(x)
const char* lrpx_build_flags(void) {
    // (every translation unit is compiled with the same flags: csrc/Makefile, build/.flags)
    return ""
#ifdef LRPX_EXPERIMENTS
           "LRPX_EXPERIMENTS"
#ifdef LRPXH_EXP
           " LRPXH_EXP=" LRPX_STR(LRPXH_EXP)
#endif
#ifdef LRPX_EPI_EXP
           " LRPX_EPI_EXP=" LRPX_STR(LRPX_EPI_EXP)
#endif
#ifdef LRPXB_EXP
           " LRPXB_EXP=" LRPX_STR(LRPXB_EXP)
#endif
#ifdef LRPXD_EXP
           " LRPXD_EXP=" LRPX_STR(LRPXD_EXP)
#endif
#ifdef LRPXH_END_SLEEP
           " LRPXH_END_SLEEP=" LRPX_STR(LRPXH_END_SLEEP)
#endif
#ifdef LRPXH_START_SKEW
           " LRPXH_START_SKEW=" LRPX_STR(LRPXH_START_SKEW)
#endif
#ifdef LRPX_STAMP
           " LRPX_STAMP"
#endif
#endif
        ;
}

}  // extern "C"
