#!/usr/bin/env python3
"""Stride-2 block entries of ResNet-18 at n frames, each alone on the GPU: the one-launch cell kernel
(asl_block_entry_s2_bf16, conv1 + downsample) against the two gather-GEMM launches it replaces.
python tools/time_entry_s2.py [frames] [--no-ds]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv, pack_entry_conv, pack_entry_ds  # noqa: E402


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 640
    with_ds = "--no-ds" not in sys.argv
    for h, cin, cout in ((56, 64, 128), (28, 128, 256), (14, 256, 512)):
        g = torch.Generator().manual_seed(h)
        w3 = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        w1 = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
        sc, bi = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda() * 0.1
        x = torch.randn(n, h + 2, h + 2, cin, device="cuda").to(torch.bfloat16)
        x2 = torch.randn(n, (h + 2) // 2, (h + 2) // 2, 4, cin, device="cuda").to(torch.bfloat16)
        oh = h // 2
        y = torch.zeros(n, oh + 2, oh + 2, cout, dtype=torch.bfloat16, device="cuda")
        yd = torch.zeros(n, oh + 2, oh + 2, cout, dtype=torch.bfloat16, device="cuda")
        pe, pd = pack_entry_conv(w3, "cuda", sc, bi), pack_entry_ds(w1, "cuda", sc, bi)
        pc, p1 = pack_conv(w3, "cuda", scale=sc, bias=bi), pack_conv(w1, "cuda", scale=sc, bias=bi)
        ep = L.EntryParams(n=n, h=h, w=h, cin=cin, cout=cout, out_halo=1, ds_halo=1, relu=1, w_rows=cout)

        def entry():
            L.check(L.lib.asl_block_entry_s2_bf16(C.byref(ep), _p(x2), _p(pe.w), _p(sc), _p(bi), _p(pd.w) if with_ds else None,
                                                  _p(sc) if with_ds else None, _p(bi) if with_ds else None, _p(y),
                                                  _p(yd) if with_ds else None, None))
        kw = dict(n=n, h=h, w=h, cin=cin, in_halo=1, cout=cout, stride=2, out_halo=1, out_ctot=cout, out_coff=0, res_halo=0,
                  res_ctot=0, res_coff=0)
        c3 = L.Conv2dParams(kh=3, kw=3, pad=1, relu=1, w_rows=pc.w_rows, **kw)
        c1 = L.Conv2dParams(kh=1, kw=1, pad=0, relu=0, w_rows=p1.w_rows, **kw)

        def conv3():
            L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(c3), _p(x), _p(pc.w), _p(sc), _p(bi), None, _p(y), None))

        def conv1():
            L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(c1), _p(x), _p(p1.w), _p(sc), _p(bi), None, _p(yd), None))
        te, t3, t1 = timed(entry), timed(conv3), timed(conv1)
        gf = 2.0 * n * oh * oh * cout * cin * (9 + (1 if with_ds else 0)) / 1e9
        print(f"{h}x{h} {cin}->{cout}, {n} frames: cell entry {te:7.1f} us ({gf / te * 1e-3:6.0f} TF/s)   "
              f"gather 3x3/s2 {t3:7.1f} + 1x1/s2 {t1:6.1f} = {t3 + (t1 if with_ds else 0):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
