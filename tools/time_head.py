#!/usr/bin/env python3
"""Times the temporal encoder + classifier head of a configuration alone (the per-frame features of a forward are kept and
the head is re-run on them): python tools/time_head.py <spec> <B> <T> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
m = AslClipModel(name, seed=0)
S = m.spec.size
clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None
feats, feats_b = m.frame_features(clips, want_bf16=True)
for _ in range(3):
    m._head(feats, feats_b, B, T, kp)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    m._head(feats, feats_b, B, T, kp)
e1.record()
torch.cuda.synchronize()
print(f"{name} B={B} T={T}: head {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per call", flush=True)
