// Diagnostic: per-workgroup phase times of the patch conv (needs a library built with -DASL_STAMPS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "asl_hip.h"
int main(int argc, char** argv) {
    const int frames = argc > 3 ? atoi(argv[3]) : 512, hw = argc > 1 ? atoi(argv[1]) : 56, c = argc > 2 ? atoi(argv[2]) : 64;
    const int hp = hw + 2;
    size_t xe = (size_t)frames * hp * hp * c, we = (size_t)((c + 127) / 128 * 128) * 9 * c;
    void *x, *y, *r, *w; unsigned long long* st;
    hipMalloc(&x, xe * 2); hipMalloc(&y, xe * 2); hipMalloc(&r, xe * 2); hipMalloc(&w, we * 2);
    hipMemset(x, 0, xe * 2); hipMemset(r, 0, xe * 2); hipMemset(w, 0, we * 2); hipMemset(y, 0, xe * 2);
    const size_t nst = 200000; hipMalloc(&st, nst * 5 * 8); hipMemset(st, 0, nst * 5 * 8);
    char buf[64]; snprintf(buf, sizeof buf, "%llx", (unsigned long long)st); setenv("ASL_STAMP_PTR", buf, 1);
    if (getenv("ASL_STAMP_NORES")) r = nullptr;
    asl_conv2d_params p = {};
    p.n = frames; p.h = p.w = hw; p.cin = c; p.in_halo = 1; p.cout = c; p.kh = p.kw = 3; p.stride = 1; p.pad = 1;
    p.out_halo = 1; p.out_ctot = c; p.res_halo = 1; p.res_ctot = c; p.relu = 1; p.w_rows = (c + 127) / 128 * 128;
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int it = 0; it < 3; ++it) { int rc = asl_conv2d_nhwc_bf16(&p, x, w, nullptr, nullptr, r, y, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipDeviceSynchronize();
    // one launch alone between events, then four back to back: the difference to the in-kernel span is launch / completion overhead
    hipEventRecord(e0); asl_conv2d_nhwc_bf16(&p, x, w, nullptr, nullptr, r, y, nullptr); hipEventRecord(e1);
    for (int it = 0; it < 4; ++it) asl_conv2d_nhwc_bf16(&p, x, w, nullptr, nullptr, r, y, nullptr);
    hipEventRecord(e2); hipDeviceSynchronize();
    float ms1, ms4; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms4, e1, e2);
    printf("events: single launch %.1f us, back-to-back %.1f us per launch\n", ms1 * 1e3, ms4 * 1e3 / 4);
    std::vector<unsigned long long> h(nst * 5);
    hipMemcpy(h.data(), st, nst * 5 * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0}; size_t n = 0; unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t b = 0; b < nst; ++b) { if (!h[b * 5]) continue; ++n; for (int k = 0; k < 4; ++k) s[k] += (double)(h[b * 5 + k + 1] - h[b * 5 + k]); t0 = std::min(t0, h[b * 5]); t1 = std::max(t1, h[b * 5 + 4]); }
    if (getenv("ASL_STAMP_HIST")) {   // start/end times relative to the first start, in stamp ticks
        std::vector<unsigned long long> starts, ends;
        for (size_t b = 0; b < nst; ++b) if (h[b * 5]) { starts.push_back(h[b * 5] - t0); ends.push_back(h[b * 5 + 4] - t0); }
        std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
        printf("starts: ");
        for (size_t i = 0; i < starts.size(); i += std::max<size_t>(1, starts.size() / 16)) printf("%llu ", starts[i]);
        printf("| last %llu\nends:   ", starts.back());
        for (size_t i = 0; i < ends.size(); i += std::max<size_t>(1, ends.size() / 16)) printf("%llu ", ends[i]);
        printf("| last %llu\n", ends.back());
    }
    printf("%dx%d c%d: %zu workgroups; mean cycles: prologue %.0f, K loop %.0f, epilogue %.0f, store drain %.0f; kernel span %.0f cycles\n", hw, hw, c, n, s[0] / n, s[1] / n, s[2] / n, s[3] / n, (double)(t1 - t0));
    return 0;
}
