#!/usr/bin/env python3
"""Host cost of issuing one forward() (time until the call returns, nothing awaited) against the
GPU time of the same forward, per batch size: shows whether a configuration is launch-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet18_mean"
    m = AslClipModel(name, seed=0)
    T, S = 64, m.spec.size
    for B in (1, 2, 4, 8, 16):
        clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
        kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None
        for _ in range(3):
            m(clips, kp)
        torch.cuda.synchronize()
        reps, host = 20, 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            h0 = time.perf_counter()
            m(clips, kp)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"{name} B={B:2d} T={T}: host issue {host / reps * 1e3:7.3f} ms, wall {wall / reps * 1e3:7.3f} ms per forward, "
              f"{B * reps / wall:8.1f} clips/s", flush=True)


if __name__ == "__main__":
    main()
