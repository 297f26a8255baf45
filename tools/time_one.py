"""Times single asl_conv2d_nhwc_bf16 launches on cuda:0.
Usage: python tools/time_one.py n h w cin cout k stride pad [res]   (halo 1 tensors, BN scale/bias, ReLU)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402
from video_asl_recognition_amd.packing import pack_conv  # noqa: E402


def run(n, h, w, cin, cout, k, stride, pad, res=0):
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn(n, h + 2, w + 2, cin, device="cuda").to(torch.bfloat16)
    y = torch.zeros(n, oh + 2, ow + 2, cout, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(n, oh + 2, ow + 2, cout, device="cuda").to(torch.bfloat16) if res else None
    wt = torch.randn(cout, cin, k, k) * 0.05
    pc = pack_conv(wt.to(torch.bfloat16).float(), "cuda", scale=torch.ones(cout).cuda(), bias=torch.zeros(cout).cuda())
    p = L.Conv2dParams(n=n, h=h, w=w, cin=cin, in_halo=1, cout=cout, kh=k, kw=k, stride=stride, pad=pad, out_halo=1,
                       out_ctot=cout, out_coff=0, res_halo=1, res_ctot=cout, res_coff=0, relu=1, w_rows=pc.w_rows)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    call = lambda: L.check(L.lib.asl_conv2d_nhwc_bf16(C.byref(p), ptr(x), ptr(pc.w), ptr(pc.scale), ptr(pc.bias), ptr(r), ptr(y), None))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * n * oh * ow * cout * cin * k * k / 1e9
    print(f"n={n} {h}x{w} c{cin}->{cout} k{k} s{stride} p{pad} res={res}: {us:8.1f} us  {gf / us / 1e3:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(*[int(v) for v in sys.argv[1:]])
    else:
        for case in [(512, 56, 56, 64, 128, 1, 2, 0), (512, 28, 28, 64, 128, 1, 1, 0), (512, 28, 28, 128, 256, 1, 2, 0),
                     (512, 14, 14, 128, 256, 1, 1, 0), (512, 56, 56, 64, 128, 3, 2, 1), (512, 28, 28, 64, 128, 3, 1, 1)]:
            run(*case)
