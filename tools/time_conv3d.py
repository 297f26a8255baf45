#!/usr/bin/env python3
"""Times single 3x3x3 / stride 1 convolution launches through asl_conv3d_ndhwc_bf16, one line per (shape, kernel variant):
python tools/time_conv3d.py [clips = 16] [variants...]   (variants: AUTO GATHER PP; SHAPE_FILTER=<substring> picks shapes)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv  # noqa: E402

# (name, d, h, cin, cout): I3D's 3x3x3 convs at T = 64, and a width sweep on its largest one
SHAPES = [
    ("2c 64->128 (sweep)", 32, 56, 64, 128),
    ("2c 64->192", 32, 56, 64, 192),
    ("2c 64->256 (sweep)", 32, 56, 64, 256),
    ("3b 96->128", 32, 28, 96, 128),
    ("3c 128->192", 32, 28, 128, 192),
    ("4b 96->208", 16, 14, 96, 208),
    ("4c 128->224", 16, 14, 128, 224),
    ("4d 128->256", 16, 14, 128, 256),
    ("4e 160->288", 16, 14, 160, 288),
    ("4f 160->320", 16, 14, 160, 320),
    ("5b 160->320", 8, 7, 160, 320),
    ("5c 192->384", 8, 7, 192, 384),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    variants = sys.argv[2:] or ["AUTO", "GATHER"]
    only = os.environ.get("SHAPE_FILTER", "")
    g = torch.Generator().manual_seed(0)
    for name, d, h, cin, cout in SHAPES:
        if only and only not in name:
            continue
        x = torch.zeros(n, d + 2, h + 2, h + 2, cin, dtype=torch.bfloat16, device="cuda")
        x[:, 1:-1, 1:-1, 1:-1] = torch.randn(n, d, h, h, cin, device="cuda").to(torch.bfloat16)
        wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (cin * 27)) ** 0.5
        pc = pack_conv(wt, "cuda", scale=torch.rand(cout).cuda() + 0.5, bias=torch.randn(cout).cuda() * 0.1)
        y = torch.zeros(n, d + 2, h + 2, h + 2, cout, dtype=torch.bfloat16, device="cuda")
        flops = 2.0 * n * d * h * h * cout * cin * 27
        line = f"{name:22s} n={n}"
        for vn in variants:
            p = L.Conv3dParams(n=n, d=d, h=h, w=h, cin=cin, in_halo_d=1, in_halo=1, cout=cout, kd=3, kh=3, kw=3, stride_d=1,
                               stride=1, pad_d=1, pad=1, out_halo_d=1, out_halo=1, out_ctot=cout, out_coff=0, relu=1,
                               w_rows=pc.w_rows, variant=getattr(L, "CONV_" + vn))
            args = (C.byref(p), C.c_void_p(x.data_ptr()), C.c_void_p(pc.w.data_ptr()), C.c_void_p(pc.scale.data_ptr()),
                    C.c_void_p(pc.bias.data_ptr()), C.c_void_p(y.data_ptr()), None)
            rc = L.lib.asl_conv3d_ndhwc_bf16(*args)
            if rc != 0:
                line += f" | {vn}: rc={rc}"
                continue
            for _ in range(3):
                L.lib.asl_conv3d_ndhwc_bf16(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                L.lib.asl_conv3d_ndhwc_bf16(*args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            line += f" | {vn}: {us:7.1f} us {flops / us / 1e6:7.1f} TF/s"
        print(line, flush=True)
        del x, y


if __name__ == "__main__":
    main()
