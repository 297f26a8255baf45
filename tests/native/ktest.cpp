// Native GPU checker + micro-benchmark for libaslhip.so (test infrastructure, not product).
// Compares every kernel against naive host loops on small shapes, then times the
// ResNet-18 layer shapes.  Build: see tests/native/Makefile.  Run on an MI355X:
//   tests/native/ktest [check|bench|all] [frames]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "asl_hip.h"

#define HIP_CHECK(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    explicit DBuf(size_t n_) : n(n_) { HIP_CHECK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16))); }
    ~DBuf() { (void)hipFree(p); }
    void up(const std::vector<T>& h) { HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> down() const {
        std::vector<T> h(n);
        HIP_CHECK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
        return h;
    }
    void zero() { HIP_CHECK(hipMemset(p, 0, n * sizeof(T))); }
};

static std::mt19937 rng(1234);
static float urand(float lo, float hi) { return std::uniform_real_distribution<float>(lo, hi)(rng); }

static int g_fail = 0;
static void report(const char* name, double maxerr, double tol, long long bad) {
    const bool ok = bad == 0 && maxerr <= tol;
    printf("%-58s max|err| %.3e (tol %.1e) bad %lld  %s\n", name, maxerr, tol, bad, ok ? "OK" : "FAIL");
    if (!ok) ++g_fail;
    fflush(stdout);
}

// ----------------------------------------------------------------------------------------
// conv2d check: halo'd NHWC bf16, naive host reference in double
// ----------------------------------------------------------------------------------------
struct ConvCase {
    int n, h, w, cin, cout, k, stride, pad, in_halo, out_halo;
    bool res, relu, scale_bias;
    int out_ctot, out_coff;
    int sample;      // > 1: verify every sample-th interior pixel (large cases); the halo is always checked in full
};

static void check_conv(const ConvCase& c) {
    const int hp = c.h + 2 * c.in_halo, wp = c.w + 2 * c.in_halo;
    const int oh = (c.h + 2 * c.pad - c.k) / c.stride + 1, ow = (c.w + 2 * c.pad - c.k) / c.stride + 1;
    const int ohp = oh + 2 * c.out_halo, owp = ow + 2 * c.out_halo;
    const int ctot = c.out_ctot ? c.out_ctot : c.cout;
    std::vector<uint16_t> x((size_t)c.n * hp * wp * c.cin, 0);
    for (int n = 0; n < c.n; ++n)
        for (int y = 0; y < c.h; ++y)
            for (int xx = 0; xx < c.w; ++xx)
                for (int ch = 0; ch < c.cin; ++ch)
                    x[(((size_t)n * hp + y + c.in_halo) * wp + xx + c.in_halo) * c.cin + ch] = f2bf(urand(-1, 1));
    const int K = c.k * c.k * c.cin;
    const int wrows = (c.cout + 127) / 128 * 128;
    std::vector<uint16_t> w((size_t)wrows * K, 0);
    for (int o = 0; o < c.cout; ++o)
        for (int k = 0; k < K; ++k) w[(size_t)o * K + k] = f2bf(urand(-1, 1) / std::sqrt((float)K));
    std::vector<float> sc(c.cout), bi(c.cout);
    for (int o = 0; o < c.cout; ++o) { sc[o] = urand(0.5f, 1.5f); bi[o] = urand(-0.5f, 0.5f); }
    std::vector<uint16_t> res((size_t)c.n * ohp * owp * c.cout, 0);
    if (c.res)
        for (auto& r : res) r = f2bf(urand(-1, 1));
    DBuf<uint16_t> dx(x.size()), dw(w.size()), dres(res.size()), dy((size_t)c.n * ohp * owp * ctot);
    DBuf<float> dsc(c.cout), dbi(c.cout);
    dx.up(x); dw.up(w); dres.up(res); dsc.up(sc); dbi.up(bi);
    HIP_CHECK(hipMemset(dy.p, 0x7f, dy.n * 2));   // poison: halo must stay untouched
    asl_conv2d_params p = {};
    p.n = c.n; p.h = c.h; p.w = c.w; p.cin = c.cin; p.in_halo = c.in_halo;
    p.cout = c.cout; p.kh = p.kw = c.k; p.stride = c.stride; p.pad = c.pad;
    p.out_halo = c.out_halo; p.out_ctot = ctot; p.out_coff = c.out_coff;
    p.res_halo = c.out_halo; p.res_ctot = c.cout; p.res_coff = 0;
    p.relu = c.relu; p.w_rows = wrows;
    int rc = asl_conv2d_nhwc_bf16(&p, dx.p, dw.p, c.scale_bias ? dsc.p : nullptr, c.scale_bias ? dbi.p : nullptr,
                                  c.res ? dres.p : nullptr, dy.p, nullptr);
    char name[160];
    snprintf(name, sizeof name, "conv n%d %dx%d c%d->%d k%d s%d p%d hi%d ho%d%s%s%s coff%d/%d", c.n, c.h, c.w, c.cin,
             c.cout, c.k, c.stride, c.pad, c.in_halo, c.out_halo, c.res ? " +res" : "", c.relu ? " relu" : "",
             c.scale_bias ? " bn" : "", c.out_coff, ctot);
    if (rc != 0) { printf("%-58s launch rc=%d (%s) FAIL\n", name, rc, asl_strerror(rc)); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto y = dy.down();
    double maxerr = 0;
    long long bad = 0;
    {   // a second run must reproduce the first bit for bit (a data race shows up as run-to-run variation)
        rc = asl_conv2d_nhwc_bf16(&p, dx.p, dw.p, c.scale_bias ? dsc.p : nullptr, c.scale_bias ? dbi.p : nullptr,
                                  c.res ? dres.p : nullptr, dy.p, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        auto y2 = dy.down();
        for (size_t i = 0; i < y.size(); ++i) bad += y[i] != y2[i];
        if (rc != 0) ++bad;
    }
    for (int n = 0; n < c.n; ++n)
        for (int yy = 0; yy < ohp; ++yy)
            for (int xx = 0; xx < owp; ++xx)
                for (int ch = 0; ch < ctot; ++ch) {
                    const uint16_t got = y[(((size_t)n * ohp + yy) * owp + xx) * ctot + ch];
                    const int oy = yy - c.out_halo, ox = xx - c.out_halo, oc = ch - c.out_coff;
                    if (oy < 0 || oy >= oh || ox < 0 || ox >= ow || oc < 0 || oc >= c.cout) {
                        if (got != 0x7f7f) ++bad;    // wrote outside its slice
                        continue;
                    }
                    if (c.sample > 1 && (((long long)n * oh + oy) * ow + ox) % c.sample) continue;
                    double acc = 0;
                    for (int r = 0; r < c.k; ++r)
                        for (int q = 0; q < c.k; ++q) {
                            const int iy = oy * c.stride + r - c.pad + c.in_halo, ix = ox * c.stride + q - c.pad + c.in_halo;
                            const uint16_t* xp = &x[(((size_t)n * hp + iy) * wp + ix) * c.cin];
                            const uint16_t* wp_ = &w[(size_t)oc * K + (r * c.k + q) * c.cin];
                            for (int ci = 0; ci < c.cin; ++ci) acc += (double)bf2f(xp[ci]) * bf2f(wp_[ci]);
                        }
                    double v = acc;
                    if (c.scale_bias) v = v * sc[oc] + bi[oc];
                    if (c.res) v += bf2f(res[(((size_t)n * ohp + yy) * owp + xx) * c.cout + oc]);
                    if (c.relu) v = std::max(v, 0.0);
                    const double err = std::fabs((double)bf2f(got) - v);
                    const double tol = 1e-2 * std::max(1.0, std::fabs(v));   // bf16 output: 2^-8 relative
                    if (!(err <= tol)) ++bad;
                    maxerr = std::max(maxerr, err / std::max(1.0, std::fabs(v)));
                }
    report(name, maxerr, 1e-2, bad);
}

// ----------------------------------------------------------------------------------------
// stem check: packed [n][h+6][w+6][4] input, weight [64][7][8][4]
// ----------------------------------------------------------------------------------------
static void check_stem(int n, int s, int cout, int out_halo) {
    const int sp = s + 6, os = s / 2, osp = os + 2 * out_halo;
    std::vector<float> clips((size_t)n * 3 * s * s);
    for (auto& v : clips) v = bf2f(f2bf(urand(-2, 2)));
    DBuf<float> dclips(clips.size());
    dclips.up(clips);
    DBuf<uint16_t> dx((size_t)n * sp * sp * 4);
    HIP_CHECK(hipMemset(dx.p, 0x7f, dx.n * 2));
    int rc = asl_pack_clips(dclips.p, 1, n, s, 3, dx.p, nullptr);
    if (rc) { printf("pack_clips rc=%d FAIL\n", rc); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto xp = dx.down();
    {   // check the packing itself
        long long bad = 0;
        for (int i = 0; i < n; ++i)
            for (int y = 0; y < sp; ++y)
                for (int x = 0; x < sp; ++x)
                    for (int c = 0; c < 4; ++c) {
                        const int yy = y - 3, xx = x - 3;
                        uint16_t want = 0;
                        if (c < 3 && yy >= 0 && yy < s && xx >= 0 && xx < s)
                            want = f2bf(clips[(((size_t)i * 3 + c) * s + yy) * s + xx]);
                        if (xp[(((size_t)i * sp + y) * sp + x) * 4 + c] != want) ++bad;
                    }
        report("pack_clips fp32 -> packed bf16 (exact)", 0, 0, bad);
    }
    std::vector<float> w((size_t)cout * 3 * 49);
    for (auto& v : w) v = bf2f(f2bf(urand(-1, 1) / std::sqrt(147.f)));
    const int wrows = (cout + 127) / 128 * 128;
    std::vector<uint16_t> wp((size_t)wrows * 224, 0);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 7; ++r)
                for (int q = 0; q < 7; ++q) wp[(size_t)o * 224 + r * 32 + q * 4 + c] = f2bf(w[(((size_t)o * 3 + c) * 7 + r) * 7 + q]);
    std::vector<float> sc(cout), bi(cout);
    for (int o = 0; o < cout; ++o) { sc[o] = urand(0.5f, 1.5f); bi[o] = urand(-0.5f, 0.5f); }
    DBuf<uint16_t> dw(wp.size()), dy((size_t)n * osp * osp * cout);
    DBuf<float> dsc(cout), dbi(cout);
    dw.up(wp); dsc.up(sc); dbi.up(bi);
    HIP_CHECK(hipMemset(dy.p, 0x7f, dy.n * 2));
    asl_stem_params p = {n, s, s, 3, cout, 7, 7, 2, 3, out_halo, 1, wrows};
    rc = asl_stem_conv_bf16(&p, dx.p, dw.p, dsc.p, dbi.p, dy.p, nullptr);
    if (rc) { printf("stem rc=%d (%s) FAIL\n", rc, asl_strerror(rc)); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto y = dy.down();
    double maxerr = 0;
    long long bad = 0;
    for (int i = 0; i < n; ++i)
        for (int yy = 0; yy < osp; ++yy)
            for (int xx = 0; xx < osp; ++xx)
                for (int o = 0; o < cout; ++o) {
                    const uint16_t got = y[(((size_t)i * osp + yy) * osp + xx) * cout + o];
                    const int oy = yy - out_halo, ox = xx - out_halo;
                    if (oy < 0 || oy >= os || ox < 0 || ox >= os) { if (got != 0x7f7f) ++bad; continue; }
                    double acc = 0;
                    for (int c = 0; c < 3; ++c)
                        for (int r = 0; r < 7; ++r)
                            for (int q = 0; q < 7; ++q) {
                                const int iy = oy * 2 + r - 3, ix = ox * 2 + q - 3;
                                if (iy < 0 || iy >= s || ix < 0 || ix >= s) continue;
                                acc += (double)clips[(((size_t)i * 3 + c) * s + iy) * s + ix] * w[(((size_t)o * 3 + c) * 7 + r) * 7 + q];
                            }
                    double v = std::max(acc * sc[o] + bi[o], 0.0);
                    const double err = std::fabs((double)bf2f(got) - v);
                    if (!(err <= 1e-2 * std::max(1.0, std::fabs(v)))) ++bad;
                    maxerr = std::max(maxerr, err / std::max(1.0, std::fabs(v)));
                }
    char name[96];
    snprintf(name, sizeof name, "stem7x7 n%d s%d ->%d ho%d", n, s, cout, out_halo);
    report(name, maxerr, 1e-2, bad);
    if (cout != 64 || out_halo != 1) return;
    // fused stem + maxpool must equal maxpool(the stem output just checked), bit for bit
    const int ps = os / 2, psp = ps + 2;
    DBuf<uint16_t> dp((size_t)n * psp * psp * 64), dq((size_t)n * psp * psp * 64);
    HIP_CHECK(hipMemset(dp.p, 0x7f, dp.n * 2));
    HIP_CHECK(hipMemset(dq.p, 0x7f, dq.n * 2));
    // the unfused stem left poison in its halo; zero it the way the planner's torch.zeros would
    std::vector<uint16_t> yz = y;
    for (int i = 0; i < n; ++i)
        for (int yy = 0; yy < osp; ++yy)
            for (int xx = 0; xx < osp; ++xx)
                if (yy == 0 || yy == osp - 1 || xx == 0 || xx == osp - 1)
                    for (int o = 0; o < 64; ++o) yz[(((size_t)i * osp + yy) * osp + xx) * 64 + o] = 0;
    dy.up(yz);
    rc = asl_maxpool3x3s2_nhwc(dy.p, n, os, os, 64, 1, dp.p, 1, nullptr);
    asl_stem_pool_params spp = {n, s, s, 64, 1, wrows};
    if (!rc) rc = asl_stem_pool_bf16(&spp, dx.p, dw.p, dsc.p, dbi.p, dq.p, nullptr);
    if (rc) { printf("stem_pool rc=%d (%s) FAIL\n", rc, asl_strerror(rc)); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto pa = dp.down();
    auto pb = dq.down();
    bad = 0;
    for (size_t i = 0; i < pa.size(); ++i)
        if (pa[i] != pb[i]) ++bad;
    snprintf(name, sizeof name, "fused stem+maxpool n%d s%d == maxpool(stem) (exact)", n, s);
    report(name, 0, 0, bad);
}

static void check_linear(int m, int n, int k, bool f32out, bool relu, bool withres) {
    std::vector<uint16_t> x((size_t)m * k), w((size_t)((n + 127) / 128 * 128) * k, 0), res((size_t)m * n);
    for (auto& v : x) v = f2bf(urand(-1, 1));
    for (int o = 0; o < n; ++o)
        for (int i = 0; i < k; ++i) w[(size_t)o * k + i] = f2bf(urand(-1, 1) / std::sqrt((float)k));
    for (auto& v : res) v = f2bf(urand(-1, 1));
    std::vector<float> bi(n);
    for (auto& v : bi) v = urand(-0.5f, 0.5f);
    DBuf<uint16_t> dx(x.size()), dw(w.size()), dres(res.size());
    DBuf<float> dbi(n);
    DBuf<char> dy((size_t)m * n * 4);
    dx.up(x); dw.up(w); dbi.up(bi); dres.up(res);
    asl_linear_params p = {m, n, k, k, n, n, relu, f32out, 0, (n + 127) / 128 * 128};
    int rc = asl_linear_bf16(&p, dx.p, dw.p, nullptr, dbi.p, withres ? dres.p : nullptr, dy.p, nullptr);
    char name[96];
    snprintf(name, sizeof name, "linear m%d n%d k%d %s%s%s", m, n, k, f32out ? "f32" : "bf16", relu ? " relu" : "", withres ? " +res" : "");
    if (rc) { printf("%s rc=%d (%s) FAIL\n", name, rc, asl_strerror(rc)); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto yb = dy.down();
    double maxerr = 0;
    long long bad = 0;
    for (int r = 0; r < m; ++r)
        for (int o = 0; o < n; ++o) {
            double acc = bi[o];
            for (int i = 0; i < k; ++i) acc += (double)bf2f(x[(size_t)r * k + i]) * bf2f(w[(size_t)o * k + i]);
            if (withres) acc += bf2f(res[(size_t)r * n + o]);
            if (relu) acc = std::max(acc, 0.0);
            double got;
            if (f32out) { float f; memcpy(&f, &yb[((size_t)r * n + o) * 4], 4); got = f; }
            else { uint16_t b; memcpy(&b, &yb[((size_t)r * n + o) * 2], 2); got = bf2f(b); }
            const double err = std::fabs(got - acc), tol = (f32out ? 1e-4 : 1e-2) * std::max(1.0, std::fabs(acc));
            if (!(err <= tol)) ++bad;
            maxerr = std::max(maxerr, err / std::max(1.0, std::fabs(acc)));
        }
    report(name, maxerr, f32out ? 1e-4 : 1e-2, bad);
}

static void check_pools() {
    const int n = 3, h = 18, w = 14, c = 64, hi = 1, ho = 1;
    const int hp = h + 2 * hi, wp = w + 2 * hi, oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    const int ohp = oh + 2 * ho, owp = ow + 2 * ho;
    std::vector<uint16_t> x((size_t)n * hp * wp * c, 0);
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < h; ++y)
            for (int xx = 0; xx < w; ++xx)
                for (int ch = 0; ch < c; ++ch) x[(((size_t)i * hp + y + hi) * wp + xx + hi) * c + ch] = f2bf(urand(0, 4));
    DBuf<uint16_t> dx(x.size()), dy((size_t)n * ohp * owp * c);
    dx.up(x);
    HIP_CHECK(hipMemset(dy.p, 0x7f, dy.n * 2));
    int rc = asl_maxpool3x3s2_nhwc(dx.p, n, h, w, c, hi, dy.p, ho, nullptr);
    if (rc) { printf("maxpool rc=%d FAIL\n", rc); ++g_fail; return; }
    HIP_CHECK(hipDeviceSynchronize());
    auto y = dy.down();
    long long bad = 0;
    for (int i = 0; i < n; ++i)
        for (int yy = 0; yy < ohp; ++yy)
            for (int xx = 0; xx < owp; ++xx)
                for (int ch = 0; ch < c; ++ch) {
                    const uint16_t got = y[(((size_t)i * ohp + yy) * owp + xx) * c + ch];
                    const int oy = yy - ho, ox = xx - ho;
                    if (oy < 0 || oy >= oh || ox < 0 || ox >= ow) { if (got != 0x7f7f) ++bad; continue; }
                    float m = -1e30f;
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q) {
                            const int iy = oy * 2 + r - 1, ix = ox * 2 + q - 1;
                            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                            m = std::max(m, bf2f(x[(((size_t)i * hp + iy + hi) * wp + ix + hi) * c + ch]));
                        }
                    if (got != f2bf(m)) ++bad;
                }
    report("maxpool 3x3 s2 p1 (exact)", 0, 0, bad);

    // avgpool + mean_t + head
    for (int cc : {64, 512, 2048}) {
        const int fh = 7, fw = 7, halo = 1, fr = 6;
        std::vector<uint16_t> f((size_t)fr * (fh + 2) * (fw + 2) * cc, 0);
        for (int i = 0; i < fr; ++i)
            for (int yy = 0; yy < fh; ++yy)
                for (int xx = 0; xx < fw; ++xx)
                    for (int ch = 0; ch < cc; ++ch) f[(((size_t)i * (fh + 2) + yy + halo) * (fw + 2) + xx + halo) * cc + ch] = f2bf(urand(0, 4));
        DBuf<uint16_t> df(f.size()), dfb((size_t)fr * cc);
        DBuf<float> dff((size_t)fr * cc), dmean((size_t)2 * cc);
        df.up(f);
        rc = asl_avgpool_nhwc(df.p, fr, fh, fw, cc, halo, dff.p, dfb.p, cc, 0, nullptr);
        if (!rc) rc = asl_mean_t_f32(dff.p, 2, 3, cc, dmean.p, nullptr);
        if (rc) { printf("avgpool/mean rc=%d FAIL\n", rc); ++g_fail; return; }
        HIP_CHECK(hipDeviceSynchronize());
        auto ff = dff.down();
        auto mm = dmean.down();
        auto fb = dfb.down();
        double maxerr = 0;
        bad = 0;
        std::vector<double> ref((size_t)fr * cc);
        for (int i = 0; i < fr; ++i)
            for (int ch = 0; ch < cc; ++ch) {
                double s = 0;
                for (int yy = 0; yy < fh; ++yy)
                    for (int xx = 0; xx < fw; ++xx) s += bf2f(f[(((size_t)i * (fh + 2) + yy + halo) * (fw + 2) + xx + halo) * cc + ch]);
                ref[(size_t)i * cc + ch] = s / 49.0;
                maxerr = std::max(maxerr, std::fabs(ff[(size_t)i * cc + ch] - s / 49.0));
                if (fb[(size_t)i * cc + ch] != f2bf(ff[(size_t)i * cc + ch])) ++bad;
            }
        for (int g = 0; g < 2; ++g)
            for (int ch = 0; ch < cc; ++ch) {
                const double s = (ref[(size_t)(3 * g) * cc + ch] + ref[(size_t)(3 * g + 1) * cc + ch] + ref[(size_t)(3 * g + 2) * cc + ch]) / 3.0;
                maxerr = std::max(maxerr, std::fabs(mm[(size_t)g * cc + ch] - s));
            }
        char name[64];
        snprintf(name, sizeof name, "avgpool 7x7 + mean_t c%d", cc);
        report(name, maxerr, 1e-5, bad);
    }
    {
        const int m = 3, f = 512, c = 100;
        std::vector<float> x((size_t)m * f), w((size_t)c * f), b(c);
        for (auto& v : x) v = urand(-1, 1);
        for (auto& v : w) v = urand(-0.05f, 0.05f);
        for (auto& v : b) v = urand(-0.1f, 0.1f);
        DBuf<float> dx2(x.size()), dw2(w.size()), db2(c), dy2((size_t)m * c), dprob((size_t)m * 5);
        DBuf<int64_t> didx((size_t)m * 5);
        dx2.up(x); dw2.up(w); db2.up(b);
        rc = asl_head_f32(dx2.p, m, f, dw2.p, db2.p, c, dy2.p, nullptr);
        if (!rc) rc = asl_softmax_topk(dy2.p, m, c, 5, didx.p, dprob.p, nullptr);
        if (rc) { printf("head rc=%d FAIL\n", rc); ++g_fail; return; }
        HIP_CHECK(hipDeviceSynchronize());
        auto yy = dy2.down();
        auto idx = didx.down();
        auto pr = dprob.down();
        double maxerr = 0;
        bad = 0;
        for (int r = 0; r < m; ++r) {
            std::vector<double> lg(c);
            for (int o = 0; o < c; ++o) {
                double s = b[o];
                for (int i = 0; i < f; ++i) s += (double)x[(size_t)r * f + i] * w[(size_t)o * f + i];
                lg[o] = s;
                maxerr = std::max(maxerr, std::fabs(yy[(size_t)r * c + o] - s));
            }
            double mx = *std::max_element(lg.begin(), lg.end()), sum = 0;
            for (double v : lg) sum += std::exp(v - mx);
            std::vector<int> ord(c);
            for (int o = 0; o < c; ++o) ord[o] = o;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b2) { return lg[a] > lg[b2]; });
            for (int j = 0; j < 5; ++j) {
                if (idx[(size_t)r * 5 + j] != ord[j]) ++bad;
                maxerr = std::max(maxerr, std::fabs(pr[(size_t)r * 5 + j] - std::exp(lg[ord[j]] - mx) / sum));
            }
        }
        report("head_f32 + softmax_topk", maxerr, 1e-5, bad);
    }
}

// ----------------------------------------------------------------------------------------
// timing of the ResNet-18 conv shapes
// ----------------------------------------------------------------------------------------
struct BenchLayer { const char* name; int hw, cin, cout, k, stride; int count; };

static void bench(int frames) {
    const BenchLayer layers[] = {
        {"l1 3x3 64->64 @56", 56, 64, 64, 3, 1, 4},      {"l2 3x3s2 64->128 @56", 56, 64, 128, 3, 2, 1},
        {"l2 1x1s2 64->128 @56", 56, 64, 128, 1, 2, 1},  {"l2 3x3 128->128 @28", 28, 128, 128, 3, 1, 3},
        {"l3 3x3s2 128->256 @28", 28, 128, 256, 3, 2, 1}, {"l3 1x1s2 128->256 @28", 28, 128, 256, 1, 2, 1},
        {"l3 3x3 256->256 @14", 14, 256, 256, 3, 1, 3},  {"l4 3x3s2 256->512 @14", 14, 256, 512, 3, 2, 1},
        {"l4 1x1s2 256->512 @14", 14, 256, 512, 1, 2, 1}, {"l4 3x3 512->512 @7", 7, 512, 512, 3, 1, 3},
    };
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    double total_ms = 0, total_flop = 0;
    {   // stem
        const int s = 224;
        DBuf<uint16_t> dx((size_t)frames * 230 * 230 * 4), dw((size_t)128 * 224), dy((size_t)frames * 114 * 114 * 64);
        std::vector<uint16_t> hx(dx.n), hw(dw.n);
        for (auto& v : hx) v = f2bf(urand(-1, 1));
        for (auto& v : hw) v = f2bf(urand(-0.1f, 0.1f));
        dx.up(hx); dw.up(hw); dy.zero();
        asl_stem_params p = {frames, s, s, 3, 64, 7, 7, 2, 3, 1, 1, 128};
        for (int it = 0; it < 2; ++it) asl_stem_conv_bf16(&p, dx.p, dw.p, nullptr, nullptr, dy.p, nullptr);
        HIP_CHECK(hipEventRecord(e0));
        const int iters = 5;
        for (int it = 0; it < iters; ++it) asl_stem_conv_bf16(&p, dx.p, dw.p, nullptr, nullptr, dy.p, nullptr);
        HIP_CHECK(hipEventRecord(e1));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double flop = 2.0 * frames * 112 * 112 * 64 * 147;
        printf("%-28s %8.3f ms  %7.1f TFLOP/s (algorithmic)\n", "stem 7x7s2 3->64 @224", ms, flop / ms * 1e-9);
        total_ms += ms; total_flop += flop;
        DBuf<uint16_t> dp((size_t)frames * 58 * 58 * 64);
        dp.zero();
        for (int it = 0; it < 2; ++it) asl_maxpool3x3s2_nhwc(dy.p, frames, 112, 112, 64, 1, dp.p, 1, nullptr);
        HIP_CHECK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) asl_maxpool3x3s2_nhwc(dy.p, frames, 112, 112, 64, 1, dp.p, 1, nullptr);
        HIP_CHECK(hipEventRecord(e1));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double bytes = (double)frames * (114.0 * 114 + 58.0 * 58) * 64 * 2;
        printf("%-28s %8.3f ms  %7.1f GB/s\n", "maxpool 3x3s2 @112", ms, bytes / ms * 1e-6);
        total_ms += ms;
        asl_stem_pool_params sp = {frames, s, s, 64, 1, 128};
        for (int it = 0; it < 2; ++it) asl_stem_pool_bf16(&sp, dx.p, dw.p, nullptr, nullptr, dp.p, nullptr);
        HIP_CHECK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) asl_stem_pool_bf16(&sp, dx.p, dw.p, nullptr, nullptr, dp.p, nullptr);
        HIP_CHECK(hipEventRecord(e1));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("%-28s %8.3f ms  %7.1f TFLOP/s (algorithmic)  [replaces stem + maxpool]\n", "fused stem+maxpool", ms, flop / ms * 1e-9);
    }
    for (const auto& L : layers) {
        const int pad = L.k / 2, hp = L.hw + 2, o = (L.hw + 2 * pad - L.k) / L.stride + 1, op = o + 2;
        DBuf<uint16_t> dx((size_t)frames * hp * hp * L.cin), dy((size_t)frames * op * op * L.cout), dr(dy.n);
        const int K = L.k * L.k * L.cin, wrows = (L.cout + 127) / 128 * 128;
        DBuf<uint16_t> dw((size_t)wrows * K);
        std::vector<uint16_t> hx(dx.n), hw(dw.n);
        for (auto& v : hx) v = f2bf(urand(-1, 1));
        for (auto& v : hw) v = f2bf(urand(-0.05f, 0.05f));
        dx.up(hx); dw.up(hw); dy.zero(); dr.zero();
        DBuf<float> dsc(L.cout), dbi(L.cout);
        dsc.up(std::vector<float>(L.cout, 1.f)); dbi.up(std::vector<float>(L.cout, 0.1f));
        asl_conv2d_params p = {};
        p.n = frames; p.h = p.w = L.hw; p.cin = L.cin; p.in_halo = 1; p.cout = L.cout; p.kh = p.kw = L.k;
        p.stride = L.stride; p.pad = pad; p.out_halo = 1; p.out_ctot = L.cout; p.res_halo = 1; p.res_ctot = L.cout;
        p.relu = 1; p.w_rows = wrows;
        for (int it = 0; it < 2; ++it) {
            int rc = asl_conv2d_nhwc_bf16(&p, dx.p, dw.p, dsc.p, dbi.p, dr.p, dy.p, nullptr);
            if (rc) { printf("%s rc=%d\n", L.name, rc); return; }
        }
        const int iters = 5;
        HIP_CHECK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) asl_conv2d_nhwc_bf16(&p, dx.p, dw.p, dsc.p, dbi.p, dr.p, dy.p, nullptr);
        HIP_CHECK(hipEventRecord(e1));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double flop = 2.0 * frames * o * o * L.cout * K;
        printf("%-28s %8.3f ms  %7.1f TFLOP/s   x%d\n", L.name, ms, flop / ms * 1e-9, L.count);
        total_ms += ms * L.count; total_flop += flop * L.count;
    }
    printf("ResNet-18 conv stack, %d frames: %.3f ms, %.1f TFLOP/s algorithmic, %.1f frames/s\n", frames, total_ms,
           total_flop / total_ms * 1e-9, frames / total_ms * 1e3);
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "all";
    const int frames = argc > 2 ? atoi(argv[2]) : 256;
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev == 0) { fprintf(stderr, "no GPU\n"); return 2; }
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs, asl_version %d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, asl_version());
    if (mode == "check" || mode == "all") {
        const ConvCase cases[] = {
            {2, 12, 12, 64, 64, 3, 1, 1, 1, 1, false, false, false, 0, 0},
            {2, 12, 12, 64, 64, 3, 1, 1, 1, 1, true, true, true, 0, 0},
            {3, 14, 10, 64, 128, 3, 2, 1, 1, 1, false, true, true, 0, 0},
            {3, 14, 10, 64, 128, 1, 2, 0, 1, 1, false, false, true, 0, 0},
            {1, 9, 7, 128, 256, 3, 1, 1, 1, 0, true, true, true, 0, 0},
            {2, 8, 8, 32, 64, 3, 2, 1, 1, 1, false, true, true, 0, 0},
            {2, 8, 8, 96, 48, 3, 1, 1, 2, 1, false, true, true, 0, 0},
            {2, 7, 7, 512, 512, 3, 1, 1, 1, 1, true, true, true, 0, 0},
            {5, 7, 7, 256, 104, 1, 1, 0, 0, 0, false, false, true, 0, 0},
            {2, 6, 6, 64, 36, 1, 1, 0, 1, 1, false, true, true, 96, 32},
            {2, 6, 6, 160, 200, 3, 1, 1, 1, 1, false, true, true, 0, 0},
            // large-tile 3x3 kernel: partial last tiles, tiles straddling rows and images, wide halo
            {23, 14, 14, 128, 128, 3, 1, 1, 1, 1, true, true, true, 0, 0},
            {3, 28, 28, 128, 256, 3, 1, 1, 1, 1, false, true, true, 0, 0},
            {21, 7, 7, 256, 128, 3, 1, 1, 1, 0, true, false, true, 0, 0},
            {2, 10, 10, 128, 128, 3, 1, 1, 2, 1, false, true, false, 0, 0},
            {40, 14, 14, 256, 256, 3, 1, 1, 1, 1, true, true, true, 0, 0, 31},
            // persistent 64->64 kernel: more tiles than CUs, so workgroups walk several tiles through both patch buffers
            {300, 56, 56, 64, 64, 3, 1, 1, 1, 1, true, true, true, 0, 0, 97},
        };
        for (const auto& c : cases) check_conv(c);
        check_stem(2, 32, 64, 1);
        check_stem(3, 56, 64, 1);
        check_stem(1, 224, 64, 1);
        check_stem(1, 48, 32, 0);
        check_linear(5, 100, 512, true, false, false);
        check_linear(130, 2048, 1024, false, true, true);
        check_linear(64, 384, 96, true, false, true);
        check_pools();
        printf("%s: %d failing checks\n", g_fail ? "FAILED" : "PASSED", g_fail);
    }
    if (mode == "bench" || mode == "all") bench(frames);
    return g_fail ? 1 : 0;
}
