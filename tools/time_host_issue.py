import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from video_asl_recognition_amd import AslClipModel
name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for streams in (1, 2):
    m = AslClipModel(name, seed=0, streams=streams)
    S = m.spec.size
    clips = torch.randn(B, T, 3, S, S, device="cuda").to(torch.bfloat16)
    kp = torch.rand(B, T, 21, 2, device="cuda") if m.spec.keypoints else None
    for _ in range(3): m(clips, kp)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps): m(clips, kp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name} B={B} streams={streams}: host issue {(t1-t0)/reps*1e3:.2f} ms/forward, wall {(t2-t0)/reps*1e3:.2f} ms/forward")
    del m
