#!/usr/bin/env python3
"""The temporal head of a configuration launched directly and replayed from a captured HIP graph:
python tools/time_head_graph.py <spec> <B> <T> [reps]"""
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


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ref = m._head(feats, feats_b, B, T, kp).clone()
direct = timed(lambda: m._head(feats, feats_b, B, T, kp))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        m._head(feats, feats_b, B, T, kp)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = m._head(feats, feats_b, B, T, kp)
g.replay()
torch.cuda.synchronize()
print("graph output equals direct:", bool(torch.equal(out, ref)))
print(f"{name} B={B} T={T}: head direct {direct:.1f} us, graph replay {timed(g.replay):.1f} us", flush=True)
