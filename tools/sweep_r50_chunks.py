import sys, time, torch
sys.path.insert(0, "/root/repo")
from video_asl_recognition_amd import AslClipModel
name, B, T = "resnet50_bilstm", 16, 64
clips = torch.randn(B, T, 3, 224, 224, device="cuda").to(torch.bfloat16)
for chunk, streams in ((256, 2), (128, 2), (128, 4), (64, 4), (96, 3), (192, 2), (256, 1), (128, 3), (342, 3)):
    m = AslClipModel(name, seed=0, frames_per_chunk=chunk, streams=streams)
    for _ in range(3):
        m(clips)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        m(clips)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"{name} B={B} T={T} chunk={chunk} streams={streams}: {dt*1e3:.2f} ms  {B/dt:.0f} clips/s", flush=True)
    del m; torch.cuda.empty_cache()
