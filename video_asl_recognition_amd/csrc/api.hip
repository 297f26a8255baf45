// Shape-level entry points of libaslhip.so: each fills the gather-GEMM's addressing from a
// friendlier description (conv2d / packed stem / conv3d / linear) and forwards to it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asl_hip.h"

extern "C" int asl_version(void) { return 104; }

extern "C" const char* asl_strerror(int rc) {
    switch (rc) {
        case ASL_OK: return "ok";
        case ASL_EINVAL: return "asl: invalid argument (shape, alignment or bounds)";
        case ASL_ENOTSUP: return "asl: no kernel variant for this request";
        case ASL_ETOOBIG: return "asl: request exceeds the library's built-in scratch capacity";
        default: return rc > 0 ? hipGetErrorString((hipError_t)rc) : "asl: unknown error";
    }
}

extern "C" int asl_conv2d_nhwc_bf16(const asl_conv2d_params* p, const void* x, const void* w, const float* scale,
                                    const float* bias, const void* res, void* y, asl_stream_t stream) {
    if (!p) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->cin <= 0 || (p->cin & 31) || p->cout <= 0) return ASL_EINVAL;
    if (p->kh <= 0 || p->kw <= 0 || p->stride <= 0 || p->pad < 0 || p->in_halo < p->pad) return ASL_EINVAL;
    if (p->out_halo < 0 || p->out_coff < 0 || p->out_coff + p->cout > p->out_ctot) return ASL_EINVAL;
    const int oh = (p->h + 2 * p->pad - p->kh) / p->stride + 1;
    const int ow = (p->w + 2 * p->pad - p->kw) / p->stride + 1;
    if (oh <= 0 || ow <= 0) return ASL_EINVAL;
    const int wp = p->w + 2 * p->in_halo, hp = p->h + 2 * p->in_halo;
    asl_gemm_params g = {};
    g.nb = p->n; g.od = 1; g.oh = oh; g.ow = ow; g.cout = p->cout;
    g.kd = 1; g.kh = p->kh; g.kw = p->kw; g.ck = p->cin;
    g.tap_sd = 0; g.tap_sh = wp * p->cin; g.tap_sw = p->cin;
    g.in.sn = (int64_t)hp * wp * p->cin; g.in.sd = 0;
    g.in.sh = p->stride * wp * p->cin; g.in.sw = p->stride * p->cin;
    g.in.off0 = ((p->in_halo - p->pad) * wp + (p->in_halo - p->pad)) * p->cin;
    const int owp = ow + 2 * p->out_halo, ohp = oh + 2 * p->out_halo;
    g.out.sn = (int64_t)ohp * owp * p->out_ctot; g.out.sd = 0;
    g.out.sh = owp * p->out_ctot; g.out.sw = p->out_ctot;
    g.out.off0 = (p->out_halo * owp + p->out_halo) * p->out_ctot + p->out_coff;
    g.y_elems = (int64_t)p->n * g.out.sn;
    if (p->out_s2d) {
        // cells of the halo-1 output image: [n][ohp/2][owp/2][4][cout]
        if (p->out_s2d != 1 || p->out_halo != 1 || p->out_coff != 0 || p->out_ctot != p->cout || (oh & 1) || (ow & 1))
            return ASL_EINVAL;
        g.out.sn = (int64_t)(ohp / 2) * (owp / 2) * 4 * p->cout;
        g.out.sh = (owp / 2) * 4 * p->cout; g.out.sw = 4 * p->cout; g.out.sd = p->cout; g.out.off0 = 0;
        g.y_elems = (int64_t)p->n * g.out.sn;
        g.out_s2d = 1;
    }
    if (res) {
        if (p->res_halo < 0 || p->res_coff < 0 || p->res_coff + p->cout > p->res_ctot) return ASL_EINVAL;
        const int rwp = ow + 2 * p->res_halo, rhp = oh + 2 * p->res_halo;
        g.res.sn = (int64_t)rhp * rwp * p->res_ctot; g.res.sd = 0;
        g.res.sh = rwp * p->res_ctot; g.res.sw = p->res_ctot;
        g.res.off0 = (p->res_halo * rwp + p->res_halo) * p->res_ctot + p->res_coff;
        g.res_elems = (int64_t)p->n * g.res.sn;
    }
    g.relu = p->relu; g.out_f32 = 0; g.w_rows = p->w_rows; g.variant = p->variant;
    g.x_elems = (int64_t)p->n * g.in.sn;
    if (p->out_split) {
        if (p->out_split != 1 || p->out_s2d || (p->out_ctot & 15) || p->out_coff + p->cout > p->out_ctot / 2) return ASL_EINVAL;
        if (res && ((p->res_ctot & 15) || p->res_coff + p->cout > p->res_ctot / 2)) return ASL_EINVAL;
        g.split_out = 1; g.out_lo = p->out_ctot / 2; g.res_lo = res ? p->res_ctot / 2 : 0;
    }
    return asl_gather_gemm_bf16(&g, x, w, scale, bias, res, y, stream);
}

extern "C" int asl_stem_conv_bf16(const asl_stem_params* p, const void* x, const void* w, const float* scale,
                                  const float* bias, void* y, asl_stream_t stream) {
    if (!p) return ASL_EINVAL;
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->cout <= 0 || p->out_halo < 0) return ASL_EINVAL;
    if (p->kh <= 0 || p->kw <= 0 || p->kw > 8 || p->stride <= 0 || p->pad < 0 || p->in_halo < p->pad)
        return ASL_EINVAL;
    const int oh = (p->h + 2 * p->pad - p->kh) / p->stride + 1;
    const int ow = (p->w + 2 * p->pad - p->kw) / p->stride + 1;
    if (oh <= 0 || ow <= 0) return ASL_EINVAL;
    // the 8-pixel window of the last output column must end inside the (halo'd) row
    if ((ow - 1) * p->stride - p->pad + 8 > p->w + p->in_halo) return ASL_EINVAL;
    const int wp = p->w + 2 * p->in_halo, hp = p->h + 2 * p->in_halo;
    asl_gemm_params g = {};
    g.nb = p->n; g.od = 1; g.oh = oh; g.ow = ow; g.cout = p->cout;
    // one "tap" per filter row: 8 pixels x 4 channels = 32 contiguous elements (pixels >= kw and
    // channel 3 carry zero weights)
    g.kd = 1; g.kh = p->kh; g.kw = 1; g.ck = 32;
    g.tap_sd = 0; g.tap_sh = wp * 4; g.tap_sw = 0;
    g.in.sn = (int64_t)hp * wp * 4; g.in.sd = 0; g.in.sh = p->stride * wp * 4; g.in.sw = p->stride * 4;
    g.in.off0 = ((p->in_halo - p->pad) * wp + (p->in_halo - p->pad)) * 4;
    const int owp = ow + 2 * p->out_halo, ohp = oh + 2 * p->out_halo;
    if (p->out_split != 0 && p->out_split != 1) return ASL_EINVAL;
    const int octot = p->out_split ? 2 * p->cout : p->cout;      // [hi | lo] halves of a split-precision output
    g.out.sn = (int64_t)ohp * owp * octot; g.out.sd = 0; g.out.sh = owp * octot; g.out.sw = octot;
    g.out.off0 = (p->out_halo * owp + p->out_halo) * octot;
    if (p->out_split) { g.split_out = 1; g.out_lo = p->cout; }
    g.relu = p->relu; g.out_f32 = 0; g.w_rows = p->w_rows;
    g.x_elems = (int64_t)p->n * g.in.sn;
    g.y_elems = (int64_t)p->n * g.out.sn;
    return asl_gather_gemm_bf16(&g, x, w, scale, bias, nullptr, y, stream);
}

extern "C" int asl_conv3d_ndhwc_bf16(const asl_conv3d_params* p, const void* x, const void* w, const float* scale,
                                     const float* bias, void* y, asl_stream_t stream) {
    if (!p) return ASL_EINVAL;
    if (p->n <= 0 || p->d <= 0 || p->h <= 0 || p->w <= 0 || p->cin <= 0 || (p->cin & 31) || p->cout <= 0)
        return ASL_EINVAL;
    if (p->kd <= 0 || p->kh <= 0 || p->kw <= 0 || p->stride <= 0 || p->stride_d <= 0) return ASL_EINVAL;
    if (p->pad < 0 || p->pad_d < 0 || p->in_halo < p->pad || p->in_halo_d < p->pad_d) return ASL_EINVAL;
    if (p->out_halo < 0 || p->out_halo_d < 0 || p->out_coff < 0 || p->out_coff + p->cout > p->out_ctot)
        return ASL_EINVAL;
    const int od = (p->d + 2 * p->pad_d - p->kd) / p->stride_d + 1;
    const int oh = (p->h + 2 * p->pad - p->kh) / p->stride + 1;
    const int ow = (p->w + 2 * p->pad - p->kw) / p->stride + 1;
    if (od <= 0 || oh <= 0 || ow <= 0) return ASL_EINVAL;
    const int wp = p->w + 2 * p->in_halo, hp = p->h + 2 * p->in_halo, dp = p->d + 2 * p->in_halo_d;
    asl_gemm_params g = {};
    g.nb = p->n; g.od = od; g.oh = oh; g.ow = ow; g.cout = p->cout;
    g.kd = p->kd; g.kh = p->kh; g.kw = p->kw; g.ck = p->cin;
    const int64_t plane = (int64_t)hp * wp * p->cin;
    if (plane * p->stride_d > 0x7fffffffLL) return ASL_EINVAL;
    g.tap_sd = (int32_t)plane; g.tap_sh = wp * p->cin; g.tap_sw = p->cin;
    g.in.sn = plane * dp; g.in.sd = (int32_t)(p->stride_d * plane);
    g.in.sh = p->stride * wp * p->cin; g.in.sw = p->stride * p->cin;
    g.in.off0 = (int32_t)((p->in_halo_d - p->pad_d) * plane) +
                ((p->in_halo - p->pad) * wp + (p->in_halo - p->pad)) * p->cin;
    const int owp = ow + 2 * p->out_halo, ohp = oh + 2 * p->out_halo, odp = od + 2 * p->out_halo_d;
    const int64_t oplane = (int64_t)ohp * owp * p->out_ctot;
    if (oplane > 0x7fffffffLL) return ASL_EINVAL;
    g.out.sn = oplane * odp; g.out.sd = (int32_t)oplane; g.out.sh = owp * p->out_ctot; g.out.sw = p->out_ctot;
    g.out.off0 = (int32_t)(p->out_halo_d * oplane) + (p->out_halo * owp + p->out_halo) * p->out_ctot + p->out_coff;
    g.relu = p->relu; g.out_f32 = 0; g.w_rows = p->w_rows; g.variant = p->variant;
    g.x_elems = (int64_t)p->n * g.in.sn;
    g.y_elems = (int64_t)p->n * g.out.sn;
    return asl_gather_gemm_bf16(&g, x, w, scale, bias, nullptr, y, stream);
}

extern "C" int asl_linear_bf16(const asl_linear_params* p, const void* x, const void* w, const float* scale,
                               const float* bias, const void* res, void* y, asl_stream_t stream) {
    if (!p) return ASL_EINVAL;
    if (p->m <= 0 || p->n <= 0 || p->k <= 0 || (p->k & 31) || p->ldx < p->k || p->ldy < p->n) return ASL_EINVAL;
    if (res && p->ldres < p->n) return ASL_EINVAL;
    asl_gemm_params g = {};
    g.nb = p->m; g.od = g.oh = g.ow = 1; g.cout = p->n;
    g.kd = g.kh = g.kw = 1; g.ck = p->k;
    g.in.sn = p->ldx; g.out.sn = p->ldy; g.res.sn = p->ldres;
    g.relu = p->relu; g.out_f32 = p->out_f32; g.res_f32 = p->res_f32; g.w_rows = p->w_rows;
    g.x_elems = (int64_t)(p->m - 1) * p->ldx + p->k;
    g.y_elems = (int64_t)(p->m - 1) * p->ldy + p->n;
    g.res_elems = (int64_t)(p->m - 1) * p->ldres + p->n;
    return asl_gather_gemm_bf16(&g, x, w, scale, bias, res, y, stream);
}

extern "C" int asl_run_ops(const asl_op* ops, int count, const asl_stream_t* streams, int nstreams, int* failed) {
    if (failed) *failed = -1;
    if (!ops || count < 0 || !streams || nstreams <= 0) return ASL_EINVAL;
    for (int k = 0; k < count; ++k) {
        const asl_op& o = ops[k];
        int rc = ASL_EINVAL;
        if (o.stream >= 0 && o.stream < nstreams) {
            asl_stream_t s = streams[o.stream];
            void* const* p = o.ptr;
            const int32_t* i = o.i;
            switch (o.kind) {
                case ASL_OP_NOP:
                    rc = 0;
                    break;
                case ASL_OP_CONV2D:
                    rc = asl_conv2d_nhwc_bf16((const asl_conv2d_params*)o.params, p[0], p[1], (const float*)p[2],
                                              (const float*)p[3], p[4], p[5], s);
                    break;
                case ASL_OP_STEM_CONV:
                    rc = asl_stem_conv_bf16((const asl_stem_params*)o.params, p[0], p[1], (const float*)p[2],
                                            (const float*)p[3], p[4], s);
                    break;
                case ASL_OP_STEM_POOL:
                    rc = asl_stem_pool_bf16((const asl_stem_pool_params*)o.params, p[0], p[1], (const float*)p[2],
                                            (const float*)p[3], p[4], s);
                    break;
                case ASL_OP_CONV3D:
                    rc = asl_conv3d_ndhwc_bf16((const asl_conv3d_params*)o.params, p[0], p[1], (const float*)p[2],
                                               (const float*)p[3], p[4], s);
                    break;
                case ASL_OP_LINEAR:
                    rc = asl_linear_bf16((const asl_linear_params*)o.params, p[0], p[1], (const float*)p[2],
                                         (const float*)p[3], p[4], p[5], s);
                    break;
                case ASL_OP_MAXPOOL3D:
                    rc = asl_maxpool3d_ndhwc((const asl_pool3d_params*)o.params, p[0], p[1], s);
                    break;
                case ASL_OP_MAXPOOL:
                    rc = asl_maxpool3x3s2_nhwc(p[0], i[0], i[1], i[2], i[3], i[4], p[1], i[5], s);
                    break;
                case ASL_OP_MAXPOOL_SPLIT:
                    rc = asl_maxpool3x3s2_split_nhwc(p[0], i[0], i[1], i[2], i[3], i[4], p[1], i[5], s);
                    break;
                case ASL_OP_AVGPOOL:
                    rc = asl_avgpool_nhwc(p[0], i[0], i[1], i[2], i[3], i[4], (float*)p[1], p[2], i[5], i[6], s);
                    break;
                case ASL_OP_PACK_CLIPS:
                    rc = asl_pack_clips(p[0], i[0], i[1], i[2], i[3], p[1], s);
                    break;
                case ASL_OP_PREPROCESS:
                    rc = i[5] ? asl_preprocess_u8_planar((const uint8_t*)p[0], i[0], i[1], i[2], i[3], o.f, o.f + 3, p[1], s)
                              : asl_preprocess_u8((const uint8_t*)p[0], i[0], i[1], i[2], i[3], i[4], o.f, o.f + 3, p[1],
                                                  (float*)p[2], s);
                    break;
                case ASL_OP_MEAN_T:
                    rc = asl_mean_t_f32((const float*)p[0], i[0], i[1], i[2], (float*)p[1], s);
                    break;
                case ASL_OP_HEAD:
                    rc = asl_head_f32((const float*)p[0], i[0], i[1], (const float*)p[1], (const float*)p[2], i[2],
                                      (float*)p[3], s);
                    break;
                case ASL_OP_BASIC_BLOCK:
                    rc = asl_basic_block_c64_bf16((const asl_block_params*)o.params, p[0], p[1], (const float*)p[2],
                                                  (const float*)p[3], p[4], (const float*)p[5], (const float*)p[6],
                                                  p[7], s);
                    break;
                case ASL_OP_BLOCK_ENTRY:
                    rc = asl_block_entry_s2_bf16((const asl_entry_params*)o.params, p[0], p[1], (const float*)p[2],
                                                 (const float*)p[3], p[4], (const float*)p[5], (const float*)p[6],
                                                 p[7], p[8], s);
                    break;
                case ASL_OP_CONV1X1_DUAL:
                    rc = asl_conv1x1_dual_bf16((const asl_dual1x1_params*)o.params, p[0], p[1], p[2], (const float*)p[3],
                                               (const float*)p[4], (const float*)p[5], p[6], s);
                    break;
                case ASL_OP_CONV1X1_CHAIN: {
                    const asl_chain1x1_params* cp = (const asl_chain1x1_params*)o.params;
                    if (!cp || !p[2] || !p[6]) { rc = ASL_EINVAL; break; }
                    const float* sa = (const float*)p[2];
                    const float* sb = (const float*)p[6];
                    rc = asl_conv1x1_chain_bf16(cp, p[0], p[1], sa, sa + cp->cmid, p[3], p[4], p[5], sb, sb + cp->cout, p[7], s);
                    break;
                }
                default:
                    rc = ASL_EINVAL;
            }
        }
        if (rc != ASL_OK) {
            if (failed) *failed = k;
            return rc;
        }
    }
    return ASL_OK;
}
