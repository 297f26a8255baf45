#!/usr/bin/env python3
"""asl_conv1x1_chain_bf16 against the two asl_conv2d_nhwc_bf16 launches it replaces (ResNet-50's 56 x 56 stage):
python tools/time_chain.py [frames = 512] [reps = 20]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p_ = lambda t: None if t is None else C.c_void_p(t.data_ptr())
h = w = 56
cin, cmid = 64, 256
for cout in (64, 128):
    g = torch.Generator().manual_seed(cout)
    wa = torch.randn(cmid, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    wb = torch.randn(cout, cmid, 1, 1, generator=g) * (2.0 / cmid) ** 0.5
    pa = pack_conv(wa, "cuda", scale=torch.ones(cmid).cuda(), bias=torch.zeros(cmid).cuda())
    pb = pack_conv(wb, "cuda", scale=torch.ones(cout).cuda(), bias=torch.zeros(cout).cuda())
    t = torch.randn(n, h + 2, w + 2, cin, device="cuda").to(torch.bfloat16)
    r = torch.randn(n, h + 2, w + 2, cmid, device="cuda").to(torch.bfloat16)
    y = torch.zeros(n, h + 2, w + 2, cmid, dtype=torch.bfloat16, device="cuda")
    y2 = torch.zeros(n, h + 2, w + 2, cout, dtype=torch.bfloat16, device="cuda")
    p1 = L.Conv2dParams(n=n, h=h, w=w, cin=cin, in_halo=1, cout=cmid, kh=1, kw=1, stride=1, pad=0, out_halo=1, out_ctot=cmid,
                        out_coff=0, res_halo=1, res_ctot=cmid, res_coff=0, relu=1, w_rows=pa.w_rows, variant=0, out_s2d=0, out_split=0)
    p2 = L.Conv2dParams(n=n, h=h, w=w, cin=cmid, in_halo=1, cout=cout, kh=1, kw=1, stride=1, pad=0, out_halo=1, out_ctot=cout,
                        out_coff=0, res_halo=0, res_ctot=0, res_coff=0, relu=1, w_rows=pb.w_rows, variant=0, out_s2d=0, out_split=0)
    pc = L.Chain1x1Params(n=n, h=h, w=w, cin=cin, cmid=cmid, cout=cout, in_halo=1, res_halo=1, mid_halo=1, out_halo=1,
                          w_rows_a=pa.w_rows, w_rows_b=pb.w_rows)

    def two():
        L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(p1), p_(t), p_(pa.w), p_(pa.scale), p_(pa.bias), p_(r), p_(y), None))
        L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(p2), p_(y), p_(pb.w), p_(pb.scale), p_(pb.bias), None, p_(y2), None))

    def chain():
        L.check(L.lib.asl_conv1x1_chain_bf16(C.byref(pc), p_(t), p_(pa.w), p_(pa.scale), p_(pa.bias), p_(r), p_(y), p_(pb.w),
                                             p_(pb.scale), p_(pb.bias), p_(y2), None))

    for name, fn in (("two launches", two), ("chain", chain), ("two launches", two), ("chain", chain)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byts = 2 * n * h * w * (cin + 2 * cmid + cout + (cmid if name != "chain" else 0))
        print(f"64->256(+res)->{cout} @56x56, {n} frames, {name:12s}: {us:7.1f} us  {byts / us / 1e3:7.1f} GB/s moved", flush=True)
