// Kernel arguments of the LDS-resident-patch 3x3 convolution (conv3x3_patch.hip), filled by the
// gather-GEMM entry point when a request has that shape.
#pragma once
#include <hip/hip_runtime.h>

#include "asl_hip.h"
#include "common.h"

namespace asl_detail {

struct PatchArgs {
    const char* x;
    const char* w;
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int M, N;              // output pixels (n*H*W), output channels
    int H, W;              // output = input spatial size
    int Hp, Wp, in_halo;   // halo'd input geometry
    int C, nchunks;        // input channels, C / 64
    int bmv;               // valid pixel slots per 256-slot tile
    int wk;                // packed K per weight row (9*C)
    long long x_bytes, w_bytes;
    long long y_bytes, res_bytes;   // extents of the output / residual tensors (kernels that address them by buffer descriptor)
    long long out_sn, res_sn;
    int out_sh, out_sw, out_off0;
    int res_sh, res_sw, res_off0;
    int relu;
    FastDiv dOW, dOHW, dODHW;   // divisors W, H*W, H*W (od == 1)
    int tiles_m, tiles_n;
    unsigned long long* stamps;   // diagnostic builds only (ASL_STAMPS): 5 s_memtime stamps per workgroup
    int stagger_ticks;     // one-time start delay (100 MHz ticks) of the second resident workgroup set
    int variant;           // ASL_CONV_* requested by the caller (0: the library chooses)
    // 3x3x3 convolutions (conv3x3_pp's KD = 3 instance only): NDHWC input [n][D + 2*in_halo_d][Hp][Wp][C], output depth D
    int D, Dp, in_halo_d;  // depth, halo'd depth, depth halo (>= 1); M = n * D * H * W, dODHW divides by D*H*W
    int out_sd;            // element stride of one output depth plane
    int tiles_per_clip;    // tiles never cross a clip: tile = clip * tiles_per_clip + t, pixels t*bmv .. of that clip
    FastDiv dTPC;
    int out_s2d;           // 0, or the plane stride (elements) of a space-to-depth output: pixel (y, x) goes to cell
                           // ((y+1)>>1, (x+1)>>1) (strides out_sh, out_sw), plane ((y+1)&1)*2 + ((x+1)&1); conv3x3_pp only
};

// compute units of the current device (cached per device; 256 when the query fails)
int device_cus();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per launch:
// `done` is the kernel's own table of per-device flags
int ensure_dynamic_lds(const void* fn, int bytes, unsigned char* done);
long long max_patch_pixels(int n, int h, int w, int in_halo, int bmv);
int patch_conv_tile(int n, int h, int w, int in_halo);
// large-tile kernel (conv3x3_pp.hip): returns -1 when the request does not fit it
int try_launch_conv3x3_wide(PatchArgs a, int n, hipStream_t stream);
int try_launch_conv3x3_pp(PatchArgs a, int n, hipStream_t stream);
int try_launch_conv3x3x3_pp(PatchArgs a, int n, hipStream_t stream);
// persistent 64 -> 64 channel kernel (conv3x3_c64.hip): returns -1 when the request does not fit it
int try_launch_conv3x3_c64(PatchArgs a, int n, hipStream_t stream);
int launch_conv3x3_patch(PatchArgs a, hipStream_t stream);
// producer / consumer form of the fused 64-channel BasicBlock (block_c64_pair.hip): -1 when the shape does not fit it
int try_launch_block_c64_pair(const asl_block_params* p, const void* x, const void* w1, const float* scale1, const float* bias1,
                              const void* w2, const float* scale2, const float* bias2, void* y, hipStream_t stream);

}  // namespace asl_detail
