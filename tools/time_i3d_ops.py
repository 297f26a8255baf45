#!/usr/bin/env python3
"""Times every launch of the I3D plan on its own (clips per launch = 16 by default, `reps` back-to-back launches between two
events) and prints kind, shape, time, TFLOP/s of the algorithmic MACs and GB/s of the algorithmic bytes (input + output once).
Usage: python tools/time_i3d_ops.py [clips = 16] [reps = 10]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_asl_recognition_amd import AslClipModel, _lib as L  # noqa: E402
from video_asl_recognition_amd.spec import SPECS  # noqa: E402


def describe(op, n):
    p = op.__defaults__[0]
    if isinstance(p, L.Conv3dParams):
        od, oh, ow = ((p.d + 2 * p.pad_d - p.kd) // p.stride_d + 1, (p.h + 2 * p.pad - p.kh) // p.stride + 1,
                      (p.w + 2 * p.pad - p.kw) // p.stride + 1)
        macs = n * od * oh * ow * p.cout * p.cin * p.kd * p.kh * p.kw
        return (f"conv {p.kd}x{p.kh}x{p.kw} {p.cin:4d}->{p.cout:4d} @{p.d}x{p.h}x{p.w}", macs,
                2 * n * (p.d * p.h * p.w * p.cin + od * oh * ow * p.cout))
    if isinstance(p, L.Pool3dParams):
        od, oh, ow = ((p.d + 2 * p.pd - p.kd) // p.sd + 1, (p.h + 2 * p.ph - p.kh) // p.sh + 1, (p.w + 2 * p.pw - p.kw) // p.sw + 1)
        return (f"pool {p.kd}x{p.kh}x{p.kw}/{p.sd}{p.sh}{p.sw} c{p.c} @{p.d}x{p.h}x{p.w}", 0,
                2 * n * p.c * (p.d * p.h * p.w + od * oh * ow))
    if isinstance(p, L.GemmParams):
        macs = n * p.od * p.oh * p.ow * 64 * 3 * 343
        return f"stem 7x7x7/2 3->64 -> {p.od}x{p.oh}x{p.ow}", macs, 2 * n * (p.od * 2 * p.oh * 2 * p.ow * 2 * 3 + p.od * p.oh * p.ow * 64)
    return "?", 0, 0


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    spec = SPECS["i3d"]
    m = AslClipModel(spec, seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(clips, spec.frames, 3, spec.size, spec.size, generator=g, device="cuda").to(torch.bfloat16)
    m(x)
    torch.cuda.synchronize()
    plan = m._plan
    stream_t = torch.cuda.current_stream()
    stream = C.c_void_p(stream_t.cuda_stream)
    total = 0.0
    for op in plan.ops:
        for _ in range(2):
            op(clips, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream_t)
        for _ in range(reps):
            op(clips, stream)
        e1.record(stream_t)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        text, macs, byts = describe(op, clips)
        total += us
        print(f"{text:52s} {us:9.1f} us  {2 * macs / us / 1e6:7.1f} TF/s  {byts / us / 1e3:7.1f} GB/s")
    print(f"# sum {total:.1f} us for {clips} clips")


if __name__ == "__main__":
    main()
