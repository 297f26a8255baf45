"""Times one forward of every model configuration on cuda:0 (random-init weights, synthetic clips) and prints
clips/s and conv TFLOP/s where known.  Usage: python tools/time_configs.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402
from video_asl_recognition_amd.spec import SPECS  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CASES = [("tiny", 2, 8), ("resnet18_mean", 8, 32), ("resnet18_mean", 4, 64), ("resnet18_mean", 8, 64),
         ("resnet18_mean", 16, 64), ("resnet18_mean", 32, 64), ("i3d", 4, 64), ("i3d", 16, 64),
         ("resnet50_bilstm", 4, 64), ("resnet50_bilstm", 16, 64), ("twostream_transformer", 2, 128),
         ("twostream_transformer", 8, 128)]
for name, B, T in CASES:
    spec = SPECS[name]
    m = AslClipModel(spec, seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    clips = torch.randn(B, T, 3, spec.size, spec.size, generator=g, device="cuda").to(torch.bfloat16)
    kp = torch.rand(B, T, 21, 2, generator=g, device="cuda") if spec.temporal == "transformer" else None
    for _ in range(2):
        m(clips, kp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m(clips, kp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:24s} B={B} T={T} S={spec.size}: {dt * 1e3:8.2f} ms/forward  {B / dt:9.1f} clips/s  {B * T / dt:10.0f} frames/s", flush=True)
    del m
    torch.cuda.empty_cache()

# the same path entered from uint8 frames (resize + normalise fused into the stem's input packing)
spec = SPECS["resnet18_mean"]
m = AslClipModel(spec, seed=0)
frames = torch.randint(0, 256, (8, 64, 240, 320, 3), dtype=torch.uint8, device="cuda")
for _ in range(2):
    m.classify_clip(frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.classify_clip(frames)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"resnet18_mean classify_clip from u8 240x320 B=8 T=64: {dt * 1e3:8.2f} ms  {8 / dt:9.1f} clips/s", flush=True)
