"""Runs the ResNet-18 clip model several times on one input and reports the largest difference between
runs (a data race shows up as run-to-run variation).  Usage: python tools/determinism.py [frames_per_chunk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = AslClipModel("resnet18_mean", frames_per_chunk=chunk, seed=0)
g = torch.Generator().manual_seed(1)
clips = torch.randn(2, 64, 3, 224, 224, generator=g).cuda()
f0, _ = m.frame_features(clips)
f0 = f0.clone()
worst = 0.0
for r in range(6):
    f, _ = m.frame_features(clips)
    d = (f - f0).abs().max().item()
    worst = max(worst, d)
    bad = ((f - f0).abs() > 0).nonzero()
    print(f"run {r}: max|diff| {d:.3e}, differing elements {bad.shape[0]}", "first", bad[:3].tolist() if bad.shape[0] else "")
print("DETERMINISTIC" if worst == 0.0 else "NONDETERMINISTIC", "env", {k: v for k, v in os.environ.items() if k.startswith("ASL_")})
