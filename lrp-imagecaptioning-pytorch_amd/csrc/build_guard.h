// Included before any switch gets its default (conv_mfma.h, common.h).
#pragma once
// ---- release / experiment separation -----------------------------------------------------------------------------------
// Every switch that makes a kernel compute WRONG results on purpose (timing experiments that leave a phase out) or adds
// profiling side effects lives behind -DLRPX_EXPERIMENTS: without it the build refuses the switch, with it the library says
// so through lrpx_build_flags() (include/lrpx.h), which tests/test_abi.py and smoke() require to be empty.
#ifndef LRPX_EXPERIMENTS
#if defined(LRPXH_EXP) || defined(LRPX_EPI_EXP) || defined(LRPXB_EXP) || defined(LRPXD_EXP) || defined(LRPXH_END_SLEEP) || \
    defined(LRPXH_START_SKEW) || defined(LRPX_STAMP)
#error "timing-experiment / profiling switches (LRPXH_EXP, LRPX_EPI_EXP, LRPXB_EXP, LRPXD_EXP, LRPXH_END_SLEEP, LRPXH_START_SKEW, LRPX_STAMP) need -DLRPX_EXPERIMENTS: such a library reports itself through lrpx_build_flags()"
#endif
#endif

