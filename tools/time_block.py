#!/usr/bin/env python3
"""Times the fused 64-channel BasicBlock against its two-launch form: python tools/time_block.py [frames]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv  # noqa: E402


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    h = w = 56
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(n, h + 2, w + 2, 64, dtype=torch.bfloat16, device="cuda")
    x[:, 1:-1, 1:-1] = torch.randn(n, h, w, 64, device="cuda").to(torch.bfloat16)
    pcs = [pack_conv(torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5, "cuda", scale=torch.rand(64).cuda() + 0.5,
                     bias=torch.randn(64).cuda() * 0.1) for _ in range(2)]
    t = torch.zeros_like(x)
    y = torch.zeros_like(x)
    bps = {v: L.BlockParams(n=n, h=h, w=w, channels=64, in_halo=1, out_halo=1, w_rows=pcs[0].w_rows, variant=v) for v in (1, 2)}
    cp = L.Conv2dParams(n=n, h=h, w=w, cin=64, in_halo=1, cout=64, kh=3, kw=3, stride=1, pad=1, out_halo=1, out_ctot=64,
                        out_coff=0, res_halo=1, res_ctot=64, res_coff=0, relu=1, w_rows=pcs[0].w_rows)

    def fused(v):
        L.check(L.lib.asl_basic_block_c64_bf16(C.byref(bps[v]), P(x), P(pcs[0].w), P(pcs[0].scale), P(pcs[0].bias), P(pcs[1].w),
                                               P(pcs[1].scale), P(pcs[1].bias), P(y), None))

    def two():
        L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(cp), P(x), P(pcs[0].w), P(pcs[0].scale), P(pcs[0].bias), None, P(t), None))
        L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(cp), P(t), P(pcs[1].w), P(pcs[1].scale), P(pcs[1].bias), P(x), P(y), None))

    flops = 2 * 2.0 * n * h * w * 64 * 576
    solo, paired = (lambda: fused(1)), (lambda: fused(2))
    for name, f in (("two launches", two), ("fused solo", solo), ("fused paired", paired), ("two launches", two),
                    ("fused solo", solo), ("fused paired", paired)):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"{name:13s} n={n}: {us:8.1f} us per block, {flops / us / 1e6:7.1f} TFLOP/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
