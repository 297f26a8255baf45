"""Times one model configuration: python tools/time_one.py NAME B T [reps]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402
from video_asl_recognition_amd.spec import SPECS  # noqa: E402

name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
spec = SPECS[name]
m = AslClipModel(spec, seed=0)
clips = torch.randn(B, T, 3, spec.size, spec.size, device="cuda").to(torch.bfloat16)
kp = torch.rand(B, T, 21, 2, device="cuda") if spec.temporal == "transformer" else None
for _ in range(2):
    m(clips, kp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m(clips, kp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{name} B={B} T={T}: {dt * 1e3:.2f} ms/forward, {B / dt:.1f} clips/s")
