"""forward(bf16 clips) vs classify_clip(uint8 frames), both device-resident, 8 clips x 64 frames, same process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

m = AslClipModel("resnet18_mean", seed=0)
clips = torch.randn(8, 64, 3, 224, 224, device="cuda").to(torch.bfloat16)
for hw in ((224, 224), (240, 320)):
    frames = torch.randint(0, 256, (8, 64, hw[0], hw[1], 3), dtype=torch.uint8, device="cuda")
    for name, fn in (("forward bf16", lambda: m(clips)), (f"classify_clip u8 {hw[0]}x{hw[1]}", lambda: m.classify_clip(frames))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"{name:32s} {dt * 1e3:6.3f} ms  {8 / dt:8.1f} clips/s", flush=True)
