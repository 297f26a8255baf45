#!/usr/bin/env python3
"""Times single convolution launches through asl_conv2d_nhwc_bf16, one line per (shape, kernel variant):
python tools/time_conv.py [frames] [variants...]   (variants: names of _lib.CONV_*, default AUTO GATHER RING)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv  # noqa: E402

# (name, h, cin, cout, k, stride, residual)
SHAPES = [
    ("r50 l1 1x1 64->256 +res", 56, 64, 256, 1, 1, True),
    ("r50 l1 1x1 256->64", 56, 256, 64, 1, 1, False),
    ("r50 l2 1x1 256->128 (56)", 56, 256, 128, 1, 1, False),
    ("r50 l2 1x1 128->512 +res", 28, 128, 512, 1, 1, True),
    ("r50 l2 1x1 512->128", 28, 512, 128, 1, 1, False),
    ("r50 l3 1x1 256->1024 +res", 14, 256, 1024, 1, 1, True),
    ("r50 l3 1x1 1024->256", 14, 1024, 256, 1, 1, False),
    ("r50 l4 1x1 512->2048 +res", 7, 512, 2048, 1, 1, True),
    ("r50 l4 1x1 2048->512", 7, 2048, 512, 1, 1, False),
    ("r50 ds 1x1s2 256->512", 56, 256, 512, 1, 2, False),
    ("r18 ds 1x1s2 64->128", 56, 64, 128, 1, 2, False),
    ("r18 ds 1x1s2 128->256", 28, 128, 256, 1, 2, False),
    ("r18 ds 1x1s2 256->512", 14, 256, 512, 1, 2, False),
    ("r18 3x3s2 64->128", 56, 64, 128, 3, 2, False),
    ("r18 3x3s2 128->256", 28, 128, 256, 3, 2, False),
    ("r18 3x3s2 256->512", 14, 256, 512, 3, 2, False),
    ("r50 3x3s2 128->128", 56, 128, 128, 3, 2, False),
    ("r18 l1 3x3 64->64 +res", 56, 64, 64, 3, 1, True),
    ("r18 l2 3x3 128->128 +res", 28, 128, 128, 3, 1, True),
    ("r18 l3 3x3 256->256 +res", 14, 256, 256, 3, 1, True),
    ("r18 l4 3x3 512->512 +res", 7, 512, 512, 3, 1, True),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    variants = sys.argv[2:] or ["AUTO", "GATHER", "RING"]
    only = os.environ.get("SHAPE_FILTER", "")
    g = torch.Generator().manual_seed(0)
    for name, h, cin, cout, k, stride, res in SHAPES:
        if only and only not in name:
            continue
        pad = k // 2
        oh = (h + 2 * pad - k) // stride + 1
        x = torch.zeros(n, h + 2, h + 2, cin, dtype=torch.bfloat16, device="cuda")
        x[:, 1:-1, 1:-1] = torch.randn(n, h, h, cin, device="cuda").to(torch.bfloat16)
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        pc = pack_conv(wt, "cuda", scale=torch.rand(cout).cuda() + 0.5, bias=torch.randn(cout).cuda() * 0.1)
        y = torch.zeros(n, oh + 2, oh + 2, cout, dtype=torch.bfloat16, device="cuda")
        r = torch.randn(n, oh + 2, oh + 2, cout, device="cuda").to(torch.bfloat16) if res else None
        flops = 2.0 * n * oh * oh * cout * cin * k * k
        byts = 2.0 * n * (h * h * cin / (stride * stride if k == 1 else 1) + oh * oh * cout * (2 if res else 1))
        line = f"{name:28s} n={n}"
        for vn in variants:
            p = L.Conv2dParams(n=n, h=h, w=h, cin=cin, in_halo=1, cout=cout, kh=k, kw=k, stride=stride, pad=pad,
                               out_halo=1, out_ctot=cout, out_coff=0, res_halo=1, res_ctot=cout, res_coff=0, relu=1,
                               w_rows=pc.w_rows, variant=getattr(L, "CONV_" + vn))
            args = (C.byref(p), C.c_void_p(x.data_ptr()), C.c_void_p(pc.w.data_ptr()), C.c_void_p(pc.scale.data_ptr()),
                    C.c_void_p(pc.bias.data_ptr()), C.c_void_p(r.data_ptr()) if res else None, C.c_void_p(y.data_ptr()), None)
            rc = L.lib.asl_conv2d_nhwc_bf16(*args)
            if rc != 0:
                line += f" | {vn}: rc={rc}"
                continue
            for _ in range(3):
                L.lib.asl_conv2d_nhwc_bf16(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                L.lib.asl_conv2d_nhwc_bf16(*args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            line += f" | {vn}: {us:7.1f} us {flops / us / 1e6:7.1f} TF/s {byts / us / 1e6:5.2f} TB/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
