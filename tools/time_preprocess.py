"""Times asl_preprocess_u8 and asl_preprocess_u8_planar alone on cuda:0 (256 frames 240x320 u8 -> 224x224 bf16, packed
with halo 3 and as planar clips)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402

n, h0, w0, s, halo = 256, 240, 320, 224, 3
src = torch.randint(0, 256, (n, h0, w0, 3), dtype=torch.uint8, device="cuda")
dst = torch.zeros(n, s + 2 * halo, s + 2 * halo, 4, dtype=torch.bfloat16, device="cuda")
mean = (C.c_float * 3)(0.485, 0.456, 0.406)
std = (C.c_float * 3)(0.229, 0.224, 0.225)
p = lambda t: C.c_void_p(t.data_ptr())
planar = torch.zeros(n, 3, s, s, dtype=torch.bfloat16, device="cuda")
calls = {"packed": (lambda: L.check(L.lib.asl_preprocess_u8(p(src), n, h0, w0, s, halo, mean, std, p(dst), None, None)), dst),
         "planar": (lambda: L.check(L.lib.asl_preprocess_u8_planar(p(src), n, h0, w0, s, mean, std, p(planar), None)), planar)}
for name, (call, out) in calls.items():
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
    print(f"preprocess ({name}) {n} frames: {us:.1f} us per call, {(src.numel() + out.numel() * 2) / us / 1e6:.2f} TB/s")
