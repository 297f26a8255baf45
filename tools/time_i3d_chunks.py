#!/usr/bin/env python3
"""I3D (and any spec) at several frames_per_chunk: clips per launch = frames_per_chunk // T.
python tools/time_i3d_chunks.py [spec] [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "i3d"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
for chunk in (256, 512, 1024, 2048):
    m = AslClipModel(name, seed=0, frames_per_chunk=chunk)
    S = m.spec.size
    clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
    kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None
    for _ in range(3):
        m(clips, kp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        m(clips, kp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"{name} B={B} T={T} frames_per_chunk={chunk}: {dt * 1e3:.2f} ms/forward, {B / dt:.1f} clips/s, workspace {m.workspace_bytes() / 2**30:.2f} GiB", flush=True)
    del m
    torch.cuda.empty_cache()
