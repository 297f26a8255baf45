"""Times asl_pack_clips alone on cuda:0 (256 frames of 224x224 bf16 -> packed halo-3 image)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L  # noqa: E402

n, s, halo = 256, 224, 3
for dt, f32 in ((torch.bfloat16, 0), (torch.float32, 1)):
    src = torch.randn(n, 3, s, s, device="cuda").to(dt)
    dst = torch.zeros(n, s + 2 * halo, s + 2 * halo, 4, dtype=torch.bfloat16, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        L.check(L.lib.asl_pack_clips(p(src), f32, n, s, halo, p(dst), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        L.check(L.lib.asl_pack_clips(p(src), f32, n, s, halo, p(dst), None))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nbytes = src.numel() * src.element_size() + dst.numel() * 2
    print(f"{dt}: {us:.1f} us per call, {nbytes / us / 1e6:.2f} TB/s")
