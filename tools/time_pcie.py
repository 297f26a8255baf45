"""PCIe-inclusive rates on cuda:0: uint8 frame batches from host memory through ClipStream (H2D of batch i+1
overlapping the kernels of batch i), and the plain host->device copy rate of the same buffers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import AslClipModel, clipio  # noqa: E402

m = AslClipModel("resnet18_mean", seed=0)
B, T, H0, W0 = 8, 64, 240, 320
batches = [torch.randint(0, 256, (B, T, H0, W0, 3), dtype=torch.uint8) for _ in range(2)]
nb = 12
seq = [batches[i % 2] for i in range(nb)]
stream = clipio.ClipStream(m, topk=5)
for _ in stream(seq[:3]):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in stream(seq):
    pass
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ClipStream u8 {H0}x{W0}: {nb * B / dt:8.1f} clips/s ({dt / nb * 1e3:.2f} ms per batch of {B}, "
      f"{batches[0].numel() / 1e6:.0f} MB per batch)")
pseq = [b.pin_memory() for b in batches]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in stream([pseq[i % 2] for i in range(nb)]):
    pass
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ClipStream, batches already pinned: {nb * B / dt:8.1f} clips/s ({dt / nb * 1e3:.2f} ms per batch)")
pin = torch.empty(batches[0].shape, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(batches[0].shape, dtype=torch.uint8, device="cuda")
pin.copy_(batches[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"pinned H2D alone: {pin.numel() / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms per batch)")
t0 = time.perf_counter()
for _ in range(5):
    pin.copy_(batches[1])
dt = (time.perf_counter() - t0) / 5
print(f"host copy into pinned staging: {pin.numel() / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms per batch)")
dframes = batches[0].cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    m.classify_clip(dframes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"classify_clip on device-resident u8 frames: {B / dt:8.1f} clips/s ({dt * 1e3:.2f} ms per batch)")
