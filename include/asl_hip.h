/*
 * asl_hip.h -- C ABI of libaslhip.so: the MI355X (gfx950) kernels behind the video-ASL
 * clip -> logits path.
 *
 * The reference snapshot defines no FFI or operator interface for this path (it holds no
 * code: /root/reference/README.md:1 is the whole README, /root/reference/.gitignore:1-5
 * the only other file).  Each entry point below therefore names the stock PyTorch CPU
 * operator(s) it stands in for -- the ops the reference's "PyTorch CPU eager" forward
 * would have called -- instead of a reference file:line.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller.  The library never
 *     allocates, frees or retains caller memory.  Its only mutable state is the per-device
 *     address of the built-in scratch array of asl_lstm_bidir_layer_static (looked up once per
 *     device); kernel selection depends on the arguments alone.
 *   - Every launcher is asynchronous on the caller-supplied stream and returns
 *     ASL_OK (0) or a negative ASL_E* code / positive hipError_t.  Geometry that a
 *     kernel cannot execute safely is rejected with ASL_EINVAL before any launch.
 *   - Activations are channels-last bf16 with an optional zero halo:
 *         x[n][d + halo_d][h + halo][w + halo][c]      (d axis only for 3-D tensors)
 *     so a padded convolution reads its zero padding from memory instead of testing
 *     bounds.  Halo cells must be zero on entry and are never written.
 *   - Physical channel counts of conv inputs are multiples of 32 (the stem's packed
 *     layout, 4 channels per pixel, is the one exception and has its own packing).
 */
#ifndef ASL_HIP_H
#define ASL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* asl_stream_t; /* == hipStream_t */

enum {
    ASL_OK = 0,
    ASL_EINVAL = -1,   /* shape / alignment / bounds violation, nothing launched */
    ASL_ENOTSUP = -2,  /* valid request that no compiled kernel variant covers   */
    ASL_ETOOBIG = -3   /* request exceeds a fixed capacity of the library (see the entry point) */
};

int asl_version(void);
const char* asl_strerror(int rc);

/* ------------------------------------------------------------------------------------
 * Addressing of one activation tensor as the gather-GEMM sees it: the element offset
 * of "output row m" (m -> image n, depth d, row h, column w of the OUTPUT grid) is
 *     n*sn + d*sd + h*sh + w*sw + off0
 * For the input tensor this is the first element the receptive field touches; for the
 * output / residual tensors it is channel 0 of the pixel.
 * ---------------------------------------------------------------------------------- */
typedef struct asl_geom {
    int64_t sn;
    int32_t sd, sh, sw;
    int32_t off0;
} asl_geom;

/* Generic gather-GEMM:  y[m][co] = act( scale[co] * sum_{tap,c} x[row(m) + tap_off(tap) + c] *
 *                                       w[co][tap*ck + c]  + bias[co] + res[m][co] )
 * Stands in for nn.Conv2d / nn.Conv3d (bias=False) + BatchNorm(eval) + residual add +
 * ReLU, and for nn.Linear (kd=kh=kw=1, od=oh=ow=1).
 *   w      packed bf16 [w_rows][wk], wk = kd*kh*kw*ck, w_rows = cout rounded up to 128,
 *          rows >= cout zero.
 *   scale  fp32 [cout] or NULL (1.0);  bias fp32 [cout] or NULL (0.0)
 *   res    residual (bf16, or fp32 if res_f32) or NULL;  y bf16 (out_f32 == 0) or fp32 (out_f32 == 1)
 *   *_elems  allocation sizes in elements, used to reject out-of-bounds geometry.       */
/* Kernel variants of the conv / GEMM entry points.  ASL_CONV_AUTO (0) lets the library choose from the
 * request's shape alone; any other value forces that kernel and returns ASL_ENOTSUP when the request does not
 * fit it (parity tests and A/B timing use this; production callers leave the field 0). */
enum {
    ASL_CONV_AUTO = 0,
    ASL_CONV_GATHER = 1,    /* generic 4-wave gather-GEMM (any shape)                                  */
    ASL_CONV_PATCH = 2,     /* 3x3/s1/p1, LDS-resident patch, 256 x 64 tile                            */
    ASL_CONV_PP = 3,        /* 3x3/s1/p1, C >= 128, N % 128 == 0: 8-wave 256 x 128 tile, two per CU    */
    ASL_CONV_PP_BIG = 4,    /* the same kernel with a 512 x 128 tile, one per CU                       */
    ASL_CONV_WIDE = 5,      /* 3x3/s1/p1, C % 64 == 0, C >= 128: 16-wave 512 x 128 tile                */
    ASL_CONV_C64 = 6        /* 3x3/s1/p1, 64 -> 64 channels: persistent, weights resident in LDS       */
};

typedef struct asl_gemm_params {
    int32_t nb, od, oh, ow;          /* output grid; M = nb*od*oh*ow            */
    int32_t cout;
    int32_t kd, kh, kw;              /* taps                                    */
    int32_t ck;                      /* contiguous K elements per tap, % 32 == 0 */
    int32_t tap_sd, tap_sh, tap_sw;  /* element offset between neighbouring taps */
    asl_geom in, out, res;
    int32_t relu;
    int32_t out_f32;
    int32_t res_f32;
    int32_t w_rows;
    int64_t x_elems, y_elems, res_elems;
    int32_t variant;                 /* ASL_CONV_*                              */
    int32_t out_s2d;                 /* 0, or 1: 2-D output written as the space-to-depth cell tensor of its halo-1 image:
                                        element offset of output pixel (h, w) = n*out.sn + ((h+1)>>1)*out.sh +
                                        ((w+1)>>1)*out.sw + (((h+1)&1)*2 + ((w+1)&1))*out.sd + out.off0 (out.sd, unused by a
                                        2-D output otherwise, is the plane stride); only the 3x3 large-tile kernel
                                        (ASL_CONV_PP) writes it: other requests return ASL_ENOTSUP */
    int32_t split_out;               /* 0, or 1: split-precision epilogue (the parity mode).  The bf16 output is written as two
                                        halves -- hi = bf16(v) at the usual place, lo = bf16(v - hi) out_lo elements further
                                        on -- and a bf16 residual is read as hi + lo (its lo half res_lo elements further
                                        on).  hi + lo carries the fp32 value to ~16 mantissa bits; a consumer reads both
                                        halves as 2 x ck input channels against weights duplicated along K.  Gather-GEMM
                                        only (other forced variants: ASL_ENOTSUP) */
    int32_t out_lo, res_lo;          /* element offsets of the lo halves (>= cout, multiples of 8) */
    int32_t reserved;                /* must be 0 */
} asl_gemm_params;

int asl_gather_gemm_bf16(const asl_gemm_params* p, const void* x, const void* w,
                         const float* scale, const float* bias, const void* res, void* y,
                         asl_stream_t stream);

/* nn.Conv2d(cin, cout, (kh,kw), stride, pad, bias=False) + BatchNorm2d(eval) [+ residual]
 * [+ ReLU] on halo'd NHWC bf16.  x: [n][h+2*in_halo][w+2*in_halo][cin]; y is written at
 * channel offset out_coff of a [n][oh+2*out_halo][ow+2*out_halo][out_ctot] tensor (so
 * branches can write slices of one concatenated tensor); res (optional) has the same
 * pixel grid as y with res_halo / res_ctot / res_coff.  Requires in_halo >= pad and
 * cin % 32 == 0.  w packed as for asl_gather_gemm_bf16 with ck = cin, tap order (kh,kw). */
typedef struct asl_conv2d_params {
    int32_t n, h, w, cin, in_halo;
    int32_t cout, kh, kw, stride, pad;
    int32_t out_halo, out_ctot, out_coff;
    int32_t res_halo, res_ctot, res_coff;
    int32_t relu;
    int32_t w_rows;
    int32_t variant;                 /* ASL_CONV_*; 0 = chosen by the library  */
    int32_t out_s2d;                 /* 1: y is the space-to-depth cell tensor of the halo-1 output (see
                                        asl_block_entry_s2_bf16); needs out_halo == 1, out_coff == 0, out_ctot == cout,
                                        even oh and ow, and a kernel that can write it (3x3/s1/p1, cin >= 128,
                                        cout % 128 == 0: ASL_ENOTSUP otherwise)            */
    int32_t out_split;               /* 1: split-precision output and residual (parity mode, see asl_gemm_params.split_out):
                                        y's hi half sits at channels [out_coff, out_coff + cout) and its lo half out_ctot / 2
                                        channels further on (out_coff + cout <= out_ctot / 2); res likewise with res_ctot / 2.
                                        The input needs no flag: its [hi | lo] halves are 2 x the logical channels, the
                                        weights are packed duplicated along K (packing.pack_conv(..., split=True)) */
} asl_conv2d_params;

int asl_conv2d_nhwc_bf16(const asl_conv2d_params* p, const void* x, const void* w,
                         const float* scale, const float* bias, const void* res, void* y,
                         asl_stream_t stream);

/* A whole ResNet BasicBlock with 64 channels in and out, fused:
 *     y = relu( bn2(conv2( relu(bn1(conv1(x))) )) + x ),   conv1, conv2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False)
 * (torchvision BasicBlock without downsample: conv1/bn1/relu/conv2/bn2, identity add, relu).  The intermediate
 * activation is rounded to bf16 exactly as the two-launch form rounds it, but never leaves the chip.
 * x: [n][h+2*in_halo][w+2*in_halo][64] (in_halo >= 1, zero halo), y: [n][h+2*out_halo][w+2*out_halo][64], y != x;
 * w1, w2 packed as for asl_conv2d_nhwc_bf16 ([w_rows][9*64]); scale/bias fp32 [64] each, all required.
 * (ASL_BLOCK_SOLO only: n * (h+2*in_halo) * (w+2*in_halo) * 128 bytes must stay below 2 GiB per call.)
 * channels other than 64: ASL_ENOTSUP.
 * variant: 0 picks the kernel; ASL_BLOCK_PAIRED = eight waves per workgroup, four running conv1 one step ahead of
 * the four running conv2 (one filter bank per wave); ASL_BLOCK_SOLO = four waves that each hold both banks and run
 * the convs one after the other.  A forced variant that does not fit the shape: ASL_ENOTSUP. */
enum { ASL_BLOCK_AUTO = 0, ASL_BLOCK_SOLO = 1, ASL_BLOCK_PAIRED = 2 };
typedef struct asl_block_params {
    int32_t n, h, w, channels;
    int32_t in_halo, out_halo;
    int32_t w_rows;
    int32_t variant;
    int32_t out_s2d;                 /* 1: y is the space-to-depth cell tensor of the halo-1 output (out_halo must be 1,
                                        h and w even; paired kernel only: ASL_ENOTSUP with ASL_BLOCK_SOLO) */
} asl_block_params;

int asl_basic_block_c64_bf16(const asl_block_params* p, const void* x, const void* w1, const float* scale1,
                             const float* bias1, const void* w2, const float* scale2, const float* bias2,
                             void* y, asl_stream_t stream);

/* Entry of a stride-2 residual block: nn.Conv2d(cin, cout, 3, stride 2, pad 1, bias=False) + BatchNorm2d(eval) [+ ReLU]
 * -> y, and, from the same launch, the block's downsample nn.Conv2d(cin, cout, 1, stride 2, bias=False) +
 * BatchNorm2d(eval) -> y_ds (torchvision BasicBlock.conv1/bn1/relu and BasicBlock.downsample of the first block of
 * layers 2-4).  Both read x2, the space-to-depth cell tensor of the block's h x w x cin input (h, w even):
 *     x2[n][(h+2)/2][(w+2)/2][4][cin] bf16, pixel (Y, X) of the halo-1 image in cell (Y>>1, X>>1), plane (Y&1)*2 + (X&1),
 * zero where the halo is; asl_conv2d_nhwc_bf16 / asl_basic_block_c64_bf16 write it with out_s2d = 1.  On cells the
 * stride-2 taps are stride-1 reads at constant offsets, so the input patch is staged once per tile and shared by all
 * taps, and the downsample (the centre tap's pixels, plane 3) needs no second pass over the input.
 *   w     bf16, MFMA-fragment order: [cout/128][cin/32][9 steps][2][4][64 lanes][8]; step order (ky,kx) = (0,0) (0,2)
 *         (2,0) (2,2) (0,1) (2,1) (1,0) (1,2) (1,1); fragment (wc, j), lane (fq*16 + fr) holds input channels
 *         32*group + 8*fq .. +7 of output channel 128*tile + 64*wc + 32*(j>>1) + 8*(fr>>2) + 4*(j&1) + (fr&3).
 *         9*cin*cout elements (cout % 128 == 0: ASL_ENOTSUP otherwise; w_rows == cout).
 *   w_ds  the same with one step per 32-channel group: [cout/128][cin/32][2][4][64][8]; NULL (with y_ds NULL): no downsample.
 *   y     [n][h/2+2*out_halo][w/2+2*out_halo][cout];  y_ds [n][h/2+2*ds_halo][w/2+2*ds_halo][cout]; interiors only.
 *   scale / bias fp32 [cout] or NULL (1 / 0).  relu applies to y only. */
typedef struct asl_entry_params {
    int32_t n, h, w, cin;
    int32_t cout;
    int32_t out_halo, ds_halo;
    int32_t relu;
    int32_t w_rows;
} asl_entry_params;

int asl_block_entry_s2_bf16(const asl_entry_params* p, const void* x2, const void* w, const float* scale,
                            const float* bias, const void* w_ds, const float* scale_ds, const float* bias_ds,
                            void* y, void* y_ds, asl_stream_t stream);

/* First-layer convolution on packed RGB: nn.Conv2d(3, cout, (kh,kw), stride, pad) [+ BatchNorm2d
 * (eval)] [+ ReLU], e.g. the ResNet stem (7x7, stride 2, pad 3).  x is the packed clip tensor
 * written by asl_pack_clips / asl_preprocess_u8: [n][h+2*in_halo][w+2*in_halo][4] bf16 with
 * channel 3 zero.  One K "tap" per filter row: 8 pixels x 4 channels = 32 contiguous elements,
 * so kw <= 8 and w packed [w_rows][kh*32] with w[co][r*32 + q*4 + c] (q >= kw, c == 3 zero).
 * Requires in_halo >= pad and in_halo >= (ow-1)*stride - pad + 8 - w so the 8-pixel window of
 * the last column stays inside the row (7x7/2/3 on even w: in_halo = 3).
 * y: [n][oh+2*out_halo][ow+2*out_halo][cout]. */
typedef struct asl_stem_params {
    int32_t n, h, w, in_halo;
    int32_t cout, kh, kw, stride, pad;
    int32_t out_halo;
    int32_t relu;
    int32_t w_rows;
    int32_t out_split;               /* 1: y has 2 * cout channels, [hi | lo] halves of the split-precision output */
} asl_stem_params;

int asl_stem_conv_bf16(const asl_stem_params* p, const void* x, const void* w,
                       const float* scale, const float* bias, void* y, asl_stream_t stream);

/* The ResNet stem fused with its pool: nn.Conv2d(3, 64, 7, stride 2, pad 3, bias=False) +
 * BatchNorm2d(eval) + ReLU + nn.MaxPool2d(3, 2, 1) in one kernel (the 112x112x64 stem activation
 * never reaches HBM).  x: packed [n][h+6][w+6][4] bf16 (asl_pack_clips / asl_preprocess_u8 with halo
 * 3); w packed as for asl_stem_conv_bf16 ([w_rows][7*32]); y: [n][h/4+2*out_halo][w/4+2*out_halo][64].
 * h, w multiples of 4; cout must be 64 (ASL_ENOTSUP otherwise).
 * planar = 1: x is the clip tensor itself, bf16 [n][3][h][w] (no packed copy, no halo: the kernel pads), and w is
 * [w_rows][6*32]: K ordered (filter row, channel, 8 taps starting one pixel left of the filter, that first tap zero),
 * 21 groups of 8 and 3 zero groups. */
typedef struct asl_stem_pool_params {
    int32_t n, h, w;
    int32_t cout, out_halo;
    int32_t w_rows;
    int32_t planar;
} asl_stem_pool_params;

int asl_stem_pool_bf16(const asl_stem_pool_params* p, const void* x, const void* w,
                       const float* scale, const float* bias, void* y, asl_stream_t stream);

/* nn.Conv3d(cin, cout, (kd,kh,kw), stride, pad=k//2, bias=False) + BatchNorm3d(eval) + ReLU
 * on halo'd NDHWC bf16: x [n][d+2*in_halo_d][h+2*in_halo][w+2*in_halo][cin]. */
typedef struct asl_conv3d_params {
    int32_t n, d, h, w, cin, in_halo_d, in_halo;
    int32_t cout, kd, kh, kw, stride_d, stride, pad_d, pad;
    int32_t out_halo_d, out_halo, out_ctot, out_coff;
    int32_t relu;
    int32_t w_rows;
    int32_t variant;                 /* ASL_CONV_AUTO, or ASL_CONV_GATHER / ASL_CONV_PP to force a kernel (3x3x3 / stride 1 / pad 1
                                        with cin >= 64 and cout % 8 == 0 runs on the 3-D instance of the large-tile kernel) */
} asl_conv3d_params;

int asl_conv3d_ndhwc_bf16(const asl_conv3d_params* p, const void* x, const void* w,
                          const float* scale, const float* bias, void* y, asl_stream_t stream);

/* The I3D stem: nn.Conv3d(3, 64, 7, stride 2, pad 3, bias=False) + BatchNorm3d(eval) + ReLU on packed RGB clips, as a
 * kernel that keeps the input rows of a run of output pixels in LDS (each input element is staged once per depth tap
 * instead of once per window).  x: packed [n][frames+6][size+6][size+6][4] bf16 (halo 3 on every axis, channel 3 zero:
 * asl_pack_clips / asl_preprocess_u8 with halo 3 into the frames behind each clip's three leading zero planes);
 * w packed [w_rows][49*32] with w[co][(kd*7 + kh)*32 + q*4 + c] (q == 7, c == 3 zero), as the gather-GEMM takes it;
 * y: [n][frames/2 + 2*out_halo_d][size/2 + 2*out_halo][size/2 + 2*out_halo][64].  frames and size even; cout must be 64
 * and every 512-pixel tile's input rows must fit 32 KiB (size <= 224 at 7x7x7 / 2), ASL_ENOTSUP otherwise -- the same
 * request then runs through asl_gather_gemm_bf16. */
typedef struct asl_stem3d_params {
    int32_t n, frames, size;
    int32_t cout;
    int32_t out_halo_d, out_halo;
    int32_t relu;
    int32_t w_rows;
} asl_stem3d_params;

int asl_stem3d_bf16(const asl_stem3d_params* p, const void* x, const void* w,
                    const float* scale, const float* bias, void* y, asl_stream_t stream);

/* The exit of a ResNet bottleneck whose shortcut is a projection, in one launch:
 *     y = relu( bn3(conv3(t)) + bn_ds(conv_ds(x)) ),   conv3: nn.Conv2d(cin, cout, 1), conv_ds: nn.Conv2d(cin2, cout, 1, stride2)
 * computed as act(scale * (W3 . t + pre_scale * (Wds . x)) + bias) with scale = bn3's scale, pre_scale = bn_ds's scale /
 * bn3's scale and bias = the two BatchNorm biases added (packing.pack_dual1x1): the shortcut tensor never reaches HBM.
 * t: [n][h+2*in_halo][w+2*in_halo][cin]; x: [n][h*stride2+2*in2_halo][w*stride2+2*in2_halo][cin2];
 * w packed [w_rows][cin2 + cin], the shortcut's columns first; cin and cin2 multiples of 64 (ASL_ENOTSUP for other
 * multiples of 32); y as for asl_conv2d_nhwc_bf16. */
typedef struct asl_dual1x1_params {
    int32_t n, h, w;
    int32_t cin, in_halo;
    int32_t cin2, in2_halo, stride2;
    int32_t cout, out_halo, out_ctot, out_coff;
    int32_t relu;
    int32_t w_rows;
} asl_dual1x1_params;

int asl_conv1x1_dual_bf16(const asl_dual1x1_params* p, const void* t, const void* x, const void* w,
                          const float* scale, const float* bias, const float* pre_scale, void* y,
                          asl_stream_t stream);

/* The exit of one ResNet bottleneck and the entry of the next, two 1x1 convolutions, in one launch:
 *     y  = relu( bn_a(conv_a(t)) + res )     conv_a: nn.Conv2d(cin, cmid, 1)   (a block's conv3 + bn3 + shortcut + ReLU)
 *     y2 = relu( bn_b(conv_b(y)) )           conv_b: nn.Conv2d(cmid, cout, 1)  (the next block's conv1 + bn1 + ReLU)
 * y is written as usual (it is the next block's shortcut) and, as bf16, stays on the chip as conv_b's operand: both
 * outputs equal those of two asl_conv2d_nhwc_bf16 calls bit for bit.  t: [n][h+2*in_halo][w+2*in_halo][cin];
 * res: [..res_halo..][cmid]; y: [..mid_halo..][cmid]; y2: [..out_halo..][cout]; weights packed as for
 * asl_conv2d_nhwc_bf16.  Built for cin = 64, cmid = 256, cout = 64 / 128 / 256 (ResNet-50's first stage and its
 * hand-over to the second); ASL_ENOTSUP for other valid sizes. */
typedef struct asl_chain1x1_params {
    int32_t n, h, w;
    int32_t cin, cmid, cout;
    int32_t in_halo, res_halo, mid_halo, out_halo;
    int32_t w_rows_a, w_rows_b;
} asl_chain1x1_params;

int asl_conv1x1_chain_bf16(const asl_chain1x1_params* p, const void* t, const void* w_a,
                           const float* scale_a, const float* bias_a, const void* res, void* y,
                           const void* w_b, const float* scale_b, const float* bias_b, void* y2,
                           asl_stream_t stream);

/* nn.Linear on bf16 rows: y[m][n] = act(scale[n] * sum_k x[m][k] w[n][k] + bias[n] (+ res)).
 * x bf16 [m][ldx] (k % 32 == 0, ldx >= k), w packed [w_rows][k], y bf16 or fp32 [m][ldy],
 * res bf16 or (res_f32) fp32 [m][ldres]. */
typedef struct asl_linear_params {
    int32_t m, n, k;
    int32_t ldx, ldy, ldres;
    int32_t relu, out_f32, res_f32;
    int32_t w_rows;
} asl_linear_params;

int asl_linear_bf16(const asl_linear_params* p, const void* x, const void* w,
                    const float* scale, const float* bias, const void* res, void* y,
                    asl_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Layout / elementwise / pooling kernels (HBM-bound)
 * ---------------------------------------------------------------------------------- */

/* clips [n][3][s][s] (fp32 if in_f32 else bf16), i.e. the [B,T,3,S,S] forward() argument
 * flattened over B*T  ->  packed first-layer input [n][s+2*halo][s+2*halo][4] bf16; halo and
 * channel 3 are zeroed by this kernel.  s must be even. */
int asl_pack_clips(const void* clips, int in_f32, int n, int s, int halo, void* y,
                   asl_stream_t stream);

/* The same for a 3-D backbone: clips [n_clips][frames][3][s][s] -> y [n_clips][frames + 2*halo_d][s+2*halo][s+2*halo][4] in
 * ONE launch, frame f of a clip into plane halo_d + f.  The 2*halo_d halo planes of every clip are not written (they must
 * already be zero); within a written plane the halo and channel 3 are zeroed as above. */
int asl_pack_clips_3d(const void* clips, int in_f32, int n_clips, int frames, int s, int halo, int halo_d,
                      void* y, asl_stream_t stream);

/* F.interpolate(bilinear, align_corners=False, antialias=False) + /255 + (x-mean)/std:
 * frames uint8 [n][h0][w0][3]  ->  packed first-layer input [n][s+2*halo][s+2*halo][4] bf16
 * (as above).  If clips_f32 != NULL the normalised fp32 [n][3][s][s] tensor is written as well. */
int asl_preprocess_u8(const uint8_t* frames, int n, int h0, int w0, int s, int halo,
                      const float mean[3], const float std_[3], void* y, float* clips_f32,
                      asl_stream_t stream);

/* The same arithmetic, written as the bf16 clip tensor [n][3][s][s] (the planar input asl_stem_pool_bf16 reads in place
 * when asl_stem_pool_params.planar is set): 6 bytes per pixel leave the kernel instead of 8 plus halo, and the stem
 * reads them once.  Values are bit-identical to the packed form's. */
int asl_preprocess_u8_planar(const uint8_t* frames, int n, int h0, int w0, int s,
                             const float mean[3], const float std_[3], void* clips_bf16,
                             asl_stream_t stream);

/* nn.MaxPool2d(3, 2, 1) on halo'd NHWC bf16 (c % 8 == 0).  Input values must be >= 0
 * wherever a window overlaps the halo (true after ReLU): the zero halo then acts as -inf. */
int asl_maxpool3x3s2_nhwc(const void* x, int n, int h, int w, int c, int in_halo,
                          void* y, int out_halo, asl_stream_t stream);

/* The same pool on a split-precision tensor (parity mode): x [n][h+2*in_halo][w+2*in_halo][2c] with channels [0, c) the hi
 * and [c, 2c) the lo halves (asl_gemm_params.split_out); the maximum is taken over hi + lo in fp32 and written as a
 * hi / lo pair into y [n][oh+2*out_halo][ow+2*out_halo][2c].  c % 8 == 0, values >= 0. */
int asl_maxpool3x3s2_split_nhwc(const void* x, int n, int h, int w, int c, int in_halo,
                                void* y, int out_halo, asl_stream_t stream);

/* nn.MaxPool3d((kd,kh,kw), (sd,sh,sw), pad = (pd,ph,pw)) on halo'd NDHWC bf16, inputs >= 0. */
typedef struct asl_pool3d_params {
    int32_t n, d, h, w, c, in_halo_d, in_halo;
    int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int32_t out_halo_d, out_halo;
} asl_pool3d_params;
int asl_maxpool3d_ndhwc(const asl_pool3d_params* p, const void* x, void* y, asl_stream_t stream);

/* x.mean((2,3)) per frame: halo'd NHWC bf16 [n][h+2*halo][w+2*halo][c] -> fp32 [n][c] and, if
 * y_bf16 != NULL, a bf16 copy with row pitch ld_bf16 for use as a GEMM operand: columns [0,c) hold
 * bf16(v); if lo_off > 0, columns [lo_off, lo_off+c) hold bf16(v - bf16(v)), so a GEMM over both
 * column blocks with the weight matrix repeated sees v to ~16 mantissa bits.
 * c/8 must be a power of two below 256 or a multiple of 256. */
int asl_avgpool_nhwc(const void* x, int n, int h, int w, int c, int halo,
                     float* y_f32, void* y_bf16, int ld_bf16, int lo_off, asl_stream_t stream);

/* y[r][c] = mean over t of x[r][t][c]  (fp32 [rows][t][c] -> fp32 [rows][c]) */
int asl_mean_t_f32(const float* x, int rows, int t, int c, float* y, asl_stream_t stream);

/* The temporal half of F.avg_pool3d(x, (window,h,w), stride 1) followed by the mean over the remaining
 * steps, on per-frame spatial means: y[r][c] = mean_{t'<=t-window} mean_{k<window} x[r][t'+k][c]. */
int asl_window_mean_t_f32(const float* x, int rows, int t, int c, int window, float* y,
                          asl_stream_t stream);

/* nn.Linear(f, c) for a handful of rows, all fp32: y[m][c] = x[m][:] . w[c][:] + b[c]. */
int asl_head_f32(const float* x, int m, int f, const float* w, const float* b, int c,
                 float* y, asl_stream_t stream);

/* F.softmax(logits, -1).topk(k): logits fp32 [m][c] -> idx int64 [m][k], prob fp32 [m][k];
 * k <= 16.  Ties resolve to the lower class index. */
int asl_softmax_topk(const float* logits, int m, int c, int k, int64_t* idx, float* prob,
                     asl_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Temporal encoders
 * ---------------------------------------------------------------------------------- */

/* One layer of nn.LSTM(bidirectional=True, batch_first=True), zero initial state, PyTorch gate
 * order (i, f, g, o).  The input projection is done beforehand by asl_linear_bf16 into
 *   gx fp32 [batch*t][ld_gx]: columns [0,4H) = x W_ih^T + b_ih + b_hh of the forward direction,
 *                             columns [4H,8H) the same for the reverse direction;
 *   w_hh bf16 [2][4H][H] (forward, reverse).
 * Writes y bf16 [batch*t][ld_y] (forward h in columns [0,H), reverse h, aligned to the input time
 * index, in [H,2H); if split_y, additionally bf16(h - bf16(h)) in columns [2H,4H), the low halves for
 * a split-precision GEMM operand) and, if ysum != NULL, ysum fp32 [batch][2H] = sum over t of ysum_scale * h (the
 * temporal mean when ysum_scale = 1/t).  state: caller scratch of 6*batch*H floats (h ping, h pong, c; zeroed here).
 * H must be a multiple of 32 (other multiples of 8 return ASL_ENOTSUP: the recurrence kernel steps K by 32; anything
 * else ASL_EINVAL).  h is carried between steps as a bf16 pair (hi + lo, ~16 mantissa bits), W_hh h runs
 * on the matrix cores with fp32 accumulation, gates and the cell state are fp32; one launch per time step on
 * `stream`. */
typedef struct asl_lstm_params {
    int32_t batch, t, hidden;
    int32_t ld_gx, ld_y;
    int32_t split_y;
    float ysum_scale;       /* each h is multiplied by this before it is added to ysum (1/t: mean) */
} asl_lstm_params;

int asl_lstm_bidir_layer(const asl_lstm_params* p, const float* gx, const void* w_hh, void* y_bf16,
                         float* ysum, float* state, asl_stream_t stream);

/* asl_lstm_bidir_layer without the caller's `state` scratch.  This entry point has no counterpart
 * in the reference (nor a torch operator it stands in for beyond nn.LSTM above): it exists for
 * callers that cannot allocate device memory around the call -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  The recurrent state lives in a device array built
 * into the library (ASL_LSTM_STATIC_STATE_FLOATS floats = 1.5 MiB, held by every GPU the library is
 * used on until the process ends); the call allocates nothing and frees nothing.
 *   - 6 * batch * hidden must not exceed ASL_LSTM_STATIC_STATE_FLOATS (e.g. batch 128 at hidden 512,
 *     batch 64 at hidden 1024); larger requests return ASL_ETOOBIG and launch nothing.
 *   - The array's address is looked up on the first call made with a given device current
 *     (hipGetSymbolAddress); make that first call outside graph capture.  Later calls consist of
 *     kernel launches only (the state is zeroed by a kernel, not by hipMemsetAsync) and can be
 *     captured into a HIP graph and replayed.
 *   - All calls on one GPU share the array: it is zeroed at the start of every call, and two calls
 *     that overlap on one GPU (different streams, or threads) race.  Serialize them.
 * Arguments, outputs and arithmetic are otherwise those of asl_lstm_bidir_layer (bit-identical). */
#define ASL_LSTM_STATIC_STATE_FLOATS (6 * 64 * 1024)
int asl_lstm_bidir_layer_static(const asl_lstm_params* p, const float* gx, const void* w_hh,
                                void* y_bf16, float* ysum, asl_stream_t stream);

/* nn.LayerNorm(d) over fp32 rows: y = (x - mean) / sqrt(var + eps) * gamma + beta, written as bf16
 * (row pitch ld_bf16) and/or fp32; at least one output.  d % 4 == 0. */
int asl_layernorm_f32(const float* x, int rows, int d, const float* gamma, const float* beta,
                      float eps, void* y_bf16, int ld_bf16, float* y_f32, asl_stream_t stream);

/* y[r][:] = x[r][:] + add[r % period][:]  (the positional-encoding add), fp32 and/or bf16 out. */
int asl_add_rows_f32(const float* x, int rows, int d, const float* add, int period, float* y_f32,
                     void* y_bf16, int ld_bf16, asl_stream_t stream);

/* fp32 [rows][k] -> bf16 [rows][ld], columns >= k zeroed (K padding of a GEMM operand). */
int asl_cast_pad_bf16(const float* x, int rows, int k, void* y_bf16, int ld, asl_stream_t stream);

/* Self-attention core of nn.MultiheadAttention (batch_first, no mask, dropout 0):
 * out[b, q, h*64:(h+1)*64] = softmax(Q_h K_h^T / sqrt(64)) V_h with Q|K|V the three heads*64-wide
 * column blocks of qkv bf16 [batch*t][ld_qkv].  t <= 128, head_dim == 64.  out bf16 [batch*t][ld_out]. */
typedef struct asl_attn_params {
    int32_t batch, t, heads, head_dim;
    int32_t ld_qkv, ld_out;
} asl_attn_params;

int asl_attention_fwd(const asl_attn_params* p, const void* qkv, void* out, asl_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Batched issue
 * ---------------------------------------------------------------------------------- */

/* One launch of one of the entry points above, as data.  asl_run_ops issues `count` of them in array order,
 * each on streams[op.stream], with the arguments, validation and return codes of the entry point the kind
 * names -- it exists so that a host written in an interpreted language pays one foreign call per batch of
 * launches instead of one per launch (a ResNet-18 chunk is 23 launches).  Nothing is retained: the array,
 * the params structs it points to and the streams array are read during the call only.
 *
 *   kind                 params                  ptr[]                                   i[] / f[]
 *   ASL_OP_NOP (0)       NULL                    -                                       (nothing is launched)
 *   ASL_OP_CONV2D        asl_conv2d_params*      x, w, scale, bias, res, y
 *   ASL_OP_STEM_CONV     asl_stem_params*        x, w, scale, bias, y
 *   ASL_OP_STEM_POOL     asl_stem_pool_params*   x, w, scale, bias, y
 *   ASL_OP_CONV3D        asl_conv3d_params*      x, w, scale, bias, y
 *   ASL_OP_LINEAR        asl_linear_params*      x, w, scale, bias, res, y
 *   ASL_OP_MAXPOOL3D     asl_pool3d_params*      x, y
 *   ASL_OP_MAXPOOL       NULL                    x, y                                    n, h, w, c, in_halo, out_halo
 *   ASL_OP_AVGPOOL       NULL                    x, y_f32, y_bf16                        n, h, w, c, halo, ld_bf16, lo_off
 *   ASL_OP_PACK_CLIPS    NULL                    clips, y                                in_f32, n, s, halo
 *   ASL_OP_PREPROCESS    NULL                    frames, y, clips_f32                    n, h0, w0, s, halo, planar / mean[3], std[3]
 *                        (planar != 0: y is the bf16 clip tensor of asl_preprocess_u8_planar; halo and clips_f32 unused)
 *   ASL_OP_MEAN_T        NULL                    x, y                                    rows, t, c
 *   ASL_OP_HEAD          NULL                    x, w, b, y                              m, f, c
 *   ASL_OP_BASIC_BLOCK   asl_block_params*       x, w1, scale1, bias1, w2, scale2, bias2, y
 *   ASL_OP_BLOCK_ENTRY   asl_entry_params*       x2, w, scale, bias, w_ds, scale_ds, bias_ds, y, y_ds
 *   ASL_OP_MAXPOOL_SPLIT NULL                    x, y                                    n, h, w, c, in_halo, out_halo
 *   ASL_OP_CONV1X1_DUAL  asl_dual1x1_params*     t, x, w, scale, bias, pre_scale, y
 *   ASL_OP_CONV1X1_CHAIN asl_chain1x1_params*    t, w_a, sb_a, res, y, w_b, sb_b, y2     (sb = scale[c] followed by bias[c],
 *                        one fp32 array of 2*c values, c = cmid for sb_a and cout for sb_b, c % 4 == 0)
 *
 * Returns ASL_OK, or the first failing op's return code with its index in *failed (if failed != NULL); ops
 * before it have been issued, ops after it have not.  ASL_EINVAL for an unknown kind, a stream index outside
 * [0, nstreams) or a missing params struct. */
enum {
    ASL_OP_NOP = 0, ASL_OP_CONV2D = 1, ASL_OP_STEM_CONV = 2, ASL_OP_STEM_POOL = 3, ASL_OP_CONV3D = 4, ASL_OP_LINEAR = 5,
    ASL_OP_MAXPOOL3D = 6, ASL_OP_MAXPOOL = 7, ASL_OP_AVGPOOL = 8, ASL_OP_PACK_CLIPS = 9, ASL_OP_PREPROCESS = 10,
    ASL_OP_MEAN_T = 11, ASL_OP_HEAD = 12, ASL_OP_BASIC_BLOCK = 13, ASL_OP_BLOCK_ENTRY = 14, ASL_OP_MAXPOOL_SPLIT = 15,
    ASL_OP_CONV1X1_DUAL = 16, ASL_OP_CONV1X1_CHAIN = 17
};

typedef struct asl_op {
    int32_t kind;
    int32_t stream;
    const void* params;
    void* ptr[10];
    int32_t i[8];
    float f[6];
} asl_op;

int asl_run_ops(const asl_op* ops, int count, const asl_stream_t* streams, int nstreams, int* failed);

#ifdef __cplusplus
}
#endif
#endif /* ASL_HIP_H */
