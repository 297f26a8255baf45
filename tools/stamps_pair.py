"""Diagnostic: prints the s_memtime stamps of the paired fused-block kernel (needs an -DASL_PAIR_STAMPS build in ASL_LIB_PATH)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import _lib as L
from video_asl_recognition_amd.packing import pack_conv
P = lambda t: C.c_void_p(t.data_ptr())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
zero = len(sys.argv) > 2
g = torch.Generator().manual_seed(0)
x = torch.zeros(n, 58, 58, 64, dtype=torch.bfloat16, device="cuda")
if not zero:
    x[:, 1:-1, 1:-1] = torch.randn(n, 56, 56, 64, device="cuda").to(torch.bfloat16)
pcs = [pack_conv(torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5 * (0 if zero else 1), "cuda", scale=torch.rand(64).cuda() + 0.5, bias=torch.randn(64).cuda() * 0.1) for _ in range(2)]
y = torch.zeros_like(x)
bp = L.BlockParams(n=n, h=56, w=56, channels=64, in_halo=1, out_halo=1, w_rows=pcs[0].w_rows, variant=2)
for _ in range(3):
    L.check(L.lib.asl_basic_block_c64_bf16(C.byref(bp), P(x), P(pcs[0].w), P(pcs[0].scale), P(pcs[0].bias), P(pcs[1].w), P(pcs[1].scale), P(pcs[1].bias), P(y), None))
torch.cuda.synchronize()
st = y.view(-1).view(torch.int64)[:808].cpu().tolist()
print("wave -> simd:", [(v >> 4) & 3 for v in st[800:808]], "cu:", [(v >> 8) & 15 for v in st[800:808]])
for role, name, per in ((0, "producer", 8), (1, "consumer", 7)):
    s = [v for v in st[400 * role:400 * role + 400]]
    t0 = s[0]
    # phases: producer: start, after dma, after carry, 4 tiles, end  (k=0 steps have 5 tiles)
    print(name, "first 70 deltas:", [s[i + 1] - s[i] for i in range(0, 70) if s[i + 1] > 0])
    last = max(i for i, v in enumerate(s) if v > 0)
    print(name, "total", s[last] - t0, "stamps", last + 1)
