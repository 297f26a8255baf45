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


def pack_conv(w: torch.Tensor, device, cin_phys: Optional[int] = None, scale=None, bias=None, split: bool = False) -> PackedConv:
    """[cout, cin, (kd,) kh, kw] -> bf16 [w_rows][taps * cin_phys], tap-major, channels innermost.
    split=True (parity mode): the input tensor carries [hi | lo] halves of every channel (2 * cin_phys channels per
    pixel), so each tap's weights are written twice, once against each half."""
    w = w.to(torch.float32)
    if w.dim() == 4:
        w = w[:, :, None]                       # kd = 1
    cout, cin, kd, kh, kw = w.shape
    cin_phys = cin_phys or _round_up(cin, 32)
    w_rows = _round_up(cout, 128)
    reps = 2 if split else 1
    p = torch.zeros(w_rows, kd, kh, kw, reps * cin_phys, dtype=torch.float32)
    for r in range(reps):
        p[:cout, :, :, :, r * cin_phys:r * cin_phys + cin] = w.permute(0, 2, 3, 4, 1)
    p = p.reshape(w_rows, kd * kh * kw * reps * cin_phys).to(torch.bfloat16).contiguous().to(device)
    return PackedConv(p, scale, bias, cout, reps * cin_phys, (kd, kh, kw), w_rows)


def pack_dual1x1(w_main: torch.Tensor, s_main, b_main, w_short: torch.Tensor, s_short, b_short, device):
    """Weights of asl_conv1x1_dual_bf16 for y = act(bn_main(conv_main(t)) + bn_short(conv_short(x))), both 1x1:
    bf16 [w_rows][cin_short + cin_main] (the shortcut's columns first), scale = the main BatchNorm's scale,
    pre_scale = shortcut scale / main scale, bias = the two biases added.  Returns (PackedConv, pre_scale), or None when a
    main-branch scale is too small to divide by (the caller keeps the two-launch form)."""
    w_main = w_main.to(torch.float32).reshape(w_main.shape[0], -1)
    w_short = w_short.to(torch.float32).reshape(w_short.shape[0], -1)
    cout, cin = w_main.shape
    cin2 = w_short.shape[1]
    s_main, s_short = s_main.to(torch.float32).cpu(), s_short.to(torch.float32).cpu()
    if cin % 64 or cin2 % 64 or bool((s_main.abs() < 1e-6 * s_short.abs().clamp_min(1e-30)).any()):
        return None
    w_rows = _round_up(cout, 128)
    p = torch.zeros(w_rows, cin2 + cin, dtype=torch.float32)
    p[:cout, :cin2] = w_short
    p[:cout, cin2:] = w_main
    pre = torch.ones(w_rows, dtype=torch.float32)
    pre[:cout] = s_short / s_main
    bias = (b_main.to(torch.float32).cpu() + b_short.to(torch.float32).cpu()).contiguous()
    pc = PackedConv(p.to(torch.bfloat16).contiguous().to(device), s_main.contiguous().to(device), bias.to(device), cout,
                    cin2 + cin, (1, 1, 1), w_rows)
    return pc, pre.contiguous().to(device)


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


def pack_stem_planar(w: torch.Tensor, device, scale=None, bias=None) -> PackedConv:
    """ResNet stem weights [cout, 3, 7, 7] for the planar-input form of the fused stem kernel: bf16 [w_rows][6 * 32],
    K ordered (filter row, channel, 8 taps): group g = 3 * row + channel holds tap slots 0..7, slot 0 (the pixel left
    of the filter) zero, slots 1..7 = taps 0..6; groups 21..23 zero."""
    w = w.to(torch.float32)
    cout, cin, kh, kw = w.shape
    assert cin == 3 and kh == 7 and kw == 7
    w_rows = _round_up(cout, 128)
    p = torch.zeros(w_rows, 24, 8, dtype=torch.float32)
    p[:cout, :21, 1:] = w.permute(0, 2, 1, 3).reshape(cout, 21, 7)      # [cout][row][ch][tap]
    p = p.reshape(w_rows, 192).to(torch.bfloat16).contiguous().to(device)
    return PackedConv(p, scale, bias, cout, 3, (1, kh, kw), w_rows)


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


# ---------------------------------------------------------------------------------------------
# Stride-2 block entry (asl_block_entry_s2_bf16): weights in MFMA-fragment order, activations as cells
# ---------------------------------------------------------------------------------------------
# K steps of one 32-channel group, as (ky, kx): plane 0 (even row, even column) four taps, plane 1 two, plane 2 two,
# plane 3 (the centre tap) one -- the order the kernel walks the planes of the cell tensor
ENTRY_TAP_ORDER = ((0, 0), (0, 2), (2, 0), (2, 2), (0, 1), (2, 1), (1, 0), (1, 2), (1, 1))


def _entry_row_order() -> torch.Tensor:
    """channel (within a 128-channel tile) of MFMA row fr of fragment (wc, j): a lane's accumulators of fragments 2h and
    2h+1 are then eight consecutive channels, which the kernel stores as one 16-byte vector"""
    idx = torch.empty(2, 4, 16, dtype=torch.long)
    for wc in range(2):
        for j in range(4):
            for fr in range(16):
                idx[wc, j, fr] = wc * 64 + (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3)
    return idx


def _entry_fragments(wk: torch.Tensor) -> torch.Tensor:
    """wk fp32 [cout, steps, 32] (per output channel: K steps of 32 input channels) -> bf16
    [cout/128][steps][wc 2][j 4][lane 64][8] with lane = fq*16 + fr holding k = 8*fq .. 8*fq+7 of row (wc, j, fr)"""
    cout, steps, _ = wk.shape
    assert cout % 128 == 0
    t = wk.reshape(cout // 128, 128, steps, 4, 8)                    # [tile][row][step][fq][8]
    rows = _entry_row_order().reshape(-1)                            # (wc, j, fr) -> row
    t = t[:, rows]                                                    # [tile][(wc j fr)][step][fq][8]
    t = t.reshape(cout // 128, 2, 4, 16, steps, 4, 8).permute(0, 4, 1, 2, 5, 3, 6)   # [tile][step][wc][j][fq][fr][8]
    return t.reshape(-1).to(torch.bfloat16).contiguous()


def pack_entry_conv(w: torch.Tensor, device, scale=None, bias=None) -> PackedConv:
    """3x3 stride-2 conv weights [cout, cin, 3, 3] (cin % 32 == 0, cout % 128 == 0) for asl_block_entry_s2_bf16"""
    w = w.to(torch.float32)
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and cin % 32 == 0 and cout % 128 == 0
    taps = torch.stack([w[:, :, ky, kx] for ky, kx in ENTRY_TAP_ORDER], 1)          # [cout][9][cin]
    wk = taps.reshape(cout, 9, cin // 32, 32).permute(0, 2, 1, 3).reshape(cout, 9 * (cin // 32), 32)   # step = 9*group + tap
    return PackedConv(_entry_fragments(wk).to(device), scale, bias, cout, cin, (1, 3, 3), cout)


def pack_entry_ds(w: torch.Tensor, device, scale=None, bias=None) -> PackedConv:
    """1x1 stride-2 downsample weights [cout, cin, 1, 1] for asl_block_entry_s2_bf16: one K step per 32-channel group"""
    w = w.to(torch.float32)
    cout, cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (1, 1) and cin % 32 == 0 and cout % 128 == 0
    wk = w.reshape(cout, cin // 32, 32)
    return PackedConv(_entry_fragments(wk).to(device), scale, bias, cout, cin, (1, 1, 1), cout)


def to_cells(x_halo: torch.Tensor) -> torch.Tensor:
    """halo-1 NHWC image [n, H+2, W+2, C] (H, W even) -> space-to-depth cell tensor [n, (H+2)/2, (W+2)/2, 4, C]:
    pixel (Y, X) in cell (Y>>1, X>>1), plane (Y&1)*2 + (X&1)"""
    n, hp, wp, c = x_halo.shape
    assert hp % 2 == 0 and wp % 2 == 0
    return x_halo.reshape(n, hp // 2, 2, wp // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, hp // 2, wp // 2, 4, c).contiguous()


def from_cells(x2: torch.Tensor) -> torch.Tensor:
    """inverse of to_cells"""
    n, hc, wc, _, c = x2.shape
    return x2.reshape(n, hc, wc, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * hc, 2 * wc, c).contiguous()
