"""Host-side weight packing: state_dict tensors -> the layouts the HIP kernels consume.

Conv weights stay exactly the bf16 values they were given (no BatchNorm folding into the
weights): BatchNorm(eval) becomes a per-channel fp32 (scale, bias) pair applied to the fp32
accumulator in the kernel epilogue, so the only rounding on the path is the bf16 store of
each layer's output.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    w: torch.Tensor                 # bf16 [w_rows][K]
    scale: Optional[torch.Tensor]   # fp32 [cout] or None
    bias: Optional[torch.Tensor]    # fp32 [cout] or None
    cout: int
    cin_phys: int
    k: tuple                        # (kd, kh, kw)
    w_rows: int


def bn_scale_bias(sd, prefix: str, eps: float, device):
    g = sd[prefix + ".weight"].to(torch.float32)
    b = sd[prefix + ".bias"].to(torch.float32)
    m = sd[prefix + ".running_mean"].to(torch.float32)
    v = sd[prefix + ".running_var"].to(torch.float32)
    scale = g / torch.sqrt(v + eps)
    bias = b - m * scale
    return scale.contiguous().to(device), bias.contiguous().to(device)


def pack_conv(w: torch.Tensor, device, cin_phys: Optional[int] = None, scale=None, bias=None) -> PackedConv:
    """[cout, cin, (kd,) kh, kw] -> bf16 [w_rows][taps * cin_phys], tap-major, channels innermost."""
    w = w.to(torch.float32)
    if w.dim() == 4:
        w = w[:, :, None]                       # kd = 1
    cout, cin, kd, kh, kw = w.shape
    cin_phys = cin_phys or _round_up(cin, 32)
    w_rows = _round_up(cout, 128)
    p = torch.zeros(w_rows, kd, kh, kw, cin_phys, dtype=torch.float32)
    p[:cout, :, :, :, :cin] = w.permute(0, 2, 3, 4, 1)
    p = p.reshape(w_rows, kd * kh * kw * cin_phys).to(torch.bfloat16).contiguous().to(device)
    return PackedConv(p, scale, bias, cout, cin_phys, (kd, kh, kw), w_rows)


def pack_stem(w: torch.Tensor, device, scale=None, bias=None) -> PackedConv:
    """First-layer weights [cout, 3, kh, kw] (kw <= 8) -> bf16 [w_rows][kh * 32]: per filter row
    8 pixels x 4 channels, zero where q >= kw or c == 3."""
    w = w.to(torch.float32)
    cout, cin, kh, kw = w.shape
    assert cin == 3 and kw <= 8
    w_rows = _round_up(cout, 128)
    p = torch.zeros(w_rows, kh, 8, 4, dtype=torch.float32)
    p[:cout, :, :kw, :3] = w.permute(0, 2, 3, 1)
    p = p.reshape(w_rows, kh * 32).to(torch.bfloat16).contiguous().to(device)
    return PackedConv(p, scale, bias, cout, 4, (1, kh, kw), w_rows)


def pack_linear(w: torch.Tensor, device, bias=None, k_phys: Optional[int] = None) -> PackedConv:
    """nn.Linear weight [n, k] -> bf16 [w_rows][k_phys] (k zero-padded to a multiple of 32)."""
    n, k = w.shape
    k_phys = k_phys or _round_up(k, 32)
    w_rows = _round_up(n, 128)
    p = torch.zeros(w_rows, k_phys, dtype=torch.float32)
    p[:n, :k] = w.to(torch.float32)
    p = p.to(torch.bfloat16).contiguous().to(device)
    b = None if bias is None else bias.to(torch.float32).contiguous().to(device)
    return PackedConv(p, None, b, n, k_phys, (1, 1, 1), w_rows)
