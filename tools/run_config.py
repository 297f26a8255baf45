#!/usr/bin/env python3
"""Runs `reps` forwards of one model configuration (for profiling under rocprofv3):
python tools/run_config.py <spec> <B> <T> [reps]      (STREAMS=<n> / CHUNK=<frames> override the model's stream count / chunk)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw = {}
if "STREAMS" in os.environ:
    kw["streams"] = int(os.environ["STREAMS"])
if "CHUNK" in os.environ:
    kw["frames_per_chunk"] = int(os.environ["CHUNK"])
m = AslClipModel(name, seed=0, **kw)
S = m.spec.size
clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None
for _ in range(2):
    m(clips, kp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m(clips, kp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{name} B={B} T={T}: {dt * 1e3:.2f} ms/forward, {B / dt:.1f} clips/s", flush=True)
