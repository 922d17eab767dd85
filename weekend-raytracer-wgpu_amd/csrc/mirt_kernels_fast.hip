// mirt_kernels_fast.hip — the OPT-IN fast-math build of the path-traced kernels (MIRT_FLAG_FAST_MATH).
//
// Same source as the exact build (mirt_kernels.hip), compiled a second time into namespace mirt::fast_build with
//   * hardware transcendentals instead of the correctly rounded sequences of mirt_device_math.h
//     (v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_sin_f32, v_cos_f32, v_exp_f32, v_log_f32: about 1 ulp each),
//   * floating-point contraction allowed and approximate division (see the Makefile's FAST_FLAGS).
// The RNG streams, the work scheduling and the exact 64-bit integer accumulation are unchanged, so an image of
// this build is deterministic too; it differs from the exact build's by a few units in the last place of some
// pixels (tests/test_gpu_fast_math.py measures the histogram).  It is never the default, no parity claim is made
// for it and no parity test uses it: it exists to show what the bit-exact arithmetic costs (bench.py `fast_math`).
#define MIRT_FAST_MATH 1
#define MIRT_KNS fast_build
#include "mirt_kernels.hip"
