import sys, time, torch
sys.path.insert(0, "/root/repo")
from video_asl_recognition_amd import AslClipModel
for B, T in ((4, 64), (8, 32), (2, 64), (1, 64)):
    clips = torch.randn(B, T, 3, 224, 224, device="cuda").to(torch.bfloat16)
    for chunk, streams in ((256, 1), (128, 2), (64, 4), (86, 3), (64, 2), (32, 2)):
        if chunk * streams > B * T * 1.01 and not (chunk, streams) == (256, 1):
            continue
        m = AslClipModel("resnet18_mean", seed=0, frames_per_chunk=chunk, streams=streams)
        for _ in range(5):
            m(clips)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            m(clips)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print(f"B={B} T={T} chunk={chunk} streams={streams}: {dt*1e3:.3f} ms  {B/dt:.0f} clips/s", flush=True)
