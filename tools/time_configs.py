#!/usr/bin/env python3
"""Times one forward of every model configuration on cuda:0 (random-init weights, synthetic clips resident in HBM)
and prints one JSON line per case with a roofline block: algorithmic conv/linear FLOPs per clip (counted by a
shape-only pass of the oracle's modules on the meta device: Conv2d/Conv3d/Linear/LSTM/attention MACs x 2, unpadded)
against the dense bf16 MFMA peak, and for ResNet-50 also the algorithmic activation bytes against the HBM peak.
Usage: python tools/time_configs.py [reps]"""
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import asl_oracle as O  # noqa: E402
from video_asl_recognition_amd import AslClipModel  # noqa: E402
from video_asl_recognition_amd.spec import SPECS  # noqa: E402

MFMA_PEAK, HBM_PEAK = 2516.6e12, 8.0e12


def algorithmic_per_clip(name, T):
    """(FLOPs, activation bytes at bf16: every conv reads its input and writes its output once, residual once)"""
    with torch.device("meta"):
        m = O.ClipModel(name, O.SPECS[name].num_classes)
    flops, byts = [0.0], [0.0]

    def hook(mod, inp, out):
        x = inp[0]
        if isinstance(mod, (nn.Conv2d, nn.Conv3d)):
            k = 1
            for d in mod.kernel_size:
                k *= d
            flops[0] += 2.0 * out.numel() * mod.in_channels * k
            byts[0] += 2.0 * (x.numel() + out.numel())
        elif isinstance(mod, nn.Linear):
            flops[0] += 2.0 * out.numel() * mod.in_features
        elif isinstance(mod, nn.LSTM):
            B, Tn, I = x.shape
            H, dirs = mod.hidden_size, 2 if mod.bidirectional else 1
            for layer in range(mod.num_layers):
                inp_dim = I if layer == 0 else H * dirs
                flops[0] += 2.0 * B * Tn * dirs * 4 * H * (inp_dim + H)
        elif isinstance(mod, nn.MultiheadAttention):
            B, Tn, E = x.shape
            flops[0] += 2.0 * B * Tn * E * 3 * E + 2.0 * B * Tn * E * E + 4.0 * B * Tn * Tn * E

    hs = [mod.register_forward_hook(hook) for mod in m.modules()
          if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.Linear, nn.LSTM, nn.MultiheadAttention))]
    S = O.SPECS[name].size
    with torch.no_grad():
        x = torch.empty(1, T, 3, S, S, device="meta")
        kp = torch.empty(1, T, 21, 2, device="meta") if name == "twostream_transformer" else None
        if name == "twostream_transformer":        # the encoder's fast path is not traceable on meta: count its parts by hand
            f = m.frame_features(x)
            kq = m.kp_mlp(kp.reshape(1, T, -1))
            z = m.fuse(torch.cat([f, kq], -1))
            d, L, ff = z.shape[-1], len(m.encoder.layers), m.encoder.layers[0].linear1.out_features
            flops[0] += L * (2.0 * T * d * 3 * d + 2.0 * T * d * d + 4.0 * T * T * d + 4.0 * T * d * ff)
            m.fc(z.mean(1))
        else:
            m(x, kp)
    for h in hs:
        h.remove()
    # residual reads: one per block output (same size as the block's last conv output); small next to the rest
    return flops[0], byts[0]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cases = [("tiny", 2, 8), ("resnet18_mean", 8, 32), ("resnet18_mean", 4, 64), ("resnet18_mean", 20, 64),
             ("i3d", 4, 64), ("i3d", 16, 64), ("resnet50_bilstm", 4, 64), ("resnet50_bilstm", 16, 64),
             ("twostream_transformer", 2, 128), ("twostream_transformer", 8, 128),
             # parity mode (hi + lo activation pairs, every conv on the gather-GEMM): what the tighter numerics cost
             ("resnet18_mean", 20, 64, True), ("resnet50_bilstm", 16, 64, True)]
    cache = {}
    for name, B, T, *rest in cases:
        parity_mode = bool(rest and rest[0])
        spec = SPECS[name]
        # 640-frame chunks for the ResNet-18 configurations (as bench.py); the model's own default otherwise
        chunk = 640 if name in ("resnet18_mean", "twostream_transformer") else None
        m = AslClipModel(spec, seed=0, frames_per_chunk=chunk, parity_mode=parity_mode)
        g = torch.Generator(device="cuda").manual_seed(0)
        clips = torch.randn(B, T, 3, spec.size, spec.size, generator=g, device="cuda").to(torch.bfloat16)
        kp = torch.rand(B, T, 21, 2, generator=g, device="cuda") if spec.temporal == "transformer" else None
        for _ in range(2):
            m(clips, kp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m(clips, kp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        if (name, T) not in cache:
            cache[(name, T)] = algorithmic_per_clip(name, T)
        fl, by = cache[(name, T)]
        tf = fl * B / dt
        rec = {"config": name, "mode": "parity" if parity_mode else "default", "clips": B, "frames": T, "size": spec.size,
               "ms_per_forward": round(dt * 1e3, 3),
               "clips_per_s": round(B / dt, 1), "algorithmic_gflop_per_clip": round(fl / 1e9, 2),
               "roofline": {"bound": "mfma", "achieved": round(tf / 1e12, 1), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                            "frac": round(tf / MFMA_PEAK, 4)},
               "hbm": {"algorithmic_mb_per_clip": round(by / 1e6, 1), "achieved_gbps": round(by * B / dt / 1e9, 1),
                       "peak_gbps": HBM_PEAK / 1e9, "frac": round(by * B / dt / HBM_PEAK, 4)}}
        print(json.dumps(rec), flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
