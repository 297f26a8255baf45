// Kernel arguments of the gather-GEMM kernel (gather_gemm.hip), filled by
// asl_gather_gemm_bf16 from the validated request.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace asl_detail {

struct GemmArgs {
    const char* x;
    const char* w;
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int M, N;
    FastDiv dOW, dOHW, dODHW;
    long long in_sn, out_sn, res_sn;
    int in_sd, in_sh, in_sw, in_off0;
    int out_sd, out_sh, out_sw, out_off0;
    int res_sd, res_sh, res_sw, res_off0;
    int KH, KW, ksteps_per_tap, ntaps;
    int tap_sd, tap_sh, tap_sw;
    int wk;                      // packed K (elements per weight row)
    long long x_bytes, w_bytes;
    int relu, out_f32, res_f32;
    int tiles_m, tiles_n;
    int out_lo, res_lo;          // split-precision output / residual (SPLIT instances): element offset of the lo half
    // DUAL instances: a second activation operand whose K steps run FIRST (a 1x1 tap: nsteps2 steps of BK contiguous
    // channels per output pixel, weights in the first nsteps2 * BK columns of every row); when they are done the
    // accumulator is multiplied by pre_scale[channel] and the primary operand's steps accumulate on top
    const char* x2;
    const float* pre_scale;
    long long in2_sn, x2_bytes;
    int in2_sh, in2_sw, in2_off0;
    int nsteps2;
};

}  // namespace asl_detail
