#!/usr/bin/env python3
"""A whole forward launched directly and replayed from a captured HIP graph (small batches, where issuing the launches
costs the host about as long as the GPU runs them): python tools/time_forward_graph.py <spec> <B> <T> [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel  # noqa: E402

name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
m = AslClipModel(name, seed=0)
S = m.spec.size
clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ref = m(clips, kp).clone()
direct = timed(lambda: m(clips, kp))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        m(clips, kp)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = m(clips, kp)
g.replay()
torch.cuda.synchronize()
print("graph output equals direct:", bool(torch.equal(out, ref)))
print(f"{name} B={B} T={T}: forward direct {direct:.3f} ms, graph replay {timed(g.replay):.3f} ms", flush=True)
