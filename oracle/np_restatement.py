"""Torch-free numpy restatement of every op family the oracle uses.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED BY THE REFERENCE (it ships no code or vectors: /root/reference/README.md:1,
/root/reference/.gitignore:1-5).  These functions restate the documented semantics of the
PyTorch CPU operators that ``oracle/asl_oracle.py`` is assembled from, in plain loops / numpy,
so ``tests/test_oracle.py`` can check the oracle's building blocks without trusting torch
twice, and so the tiny config's golden logits are checkable without torch at all.
Sizes are meant to be tiny (pure-Python loops).
"""
from __future__ import annotations

import math

import numpy as np


def conv2d(x, w, b=None, stride=1, pad=0):
    """x [N,C,H,W], w [O,C,kh,kw]: cross-correlation with zero padding (nn.Conv2d)."""
    N, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.zeros((N, C, H + 2 * pad, W + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    oh = (H + 2 * pad - kh) // stride + 1
    ow = (W + 2 * pad - kw) // stride + 1
    y = np.zeros((N, O, oh, ow), dtype=np.float64)
    for i in range(oh):
        for j in range(ow):
            patch = xp[:, :, i * stride:i * stride + kh, j * stride:j * stride + kw]
            y[:, :, i, j] = np.tensordot(patch, w.astype(np.float64), axes=([1, 2, 3], [1, 2, 3]))
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y


def conv3d(x, w, stride=1, pad=0):
    """x [N,C,D,H,W], w [O,C,kd,kh,kw] (nn.Conv3d, bias=False, symmetric padding)."""
    N, C, D, H, W = x.shape
    O, _, kd, kh, kw = w.shape
    pd, ph, pw = (pad, pad, pad) if isinstance(pad, int) else pad
    xp = np.zeros((N, C, D + 2 * pd, H + 2 * ph, W + 2 * pw), dtype=np.float64)
    xp[:, :, pd:pd + D, ph:ph + H, pw:pw + W] = x
    od, oh, ow = (D + 2 * pd - kd) // stride + 1, (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1
    y = np.zeros((N, O, od, oh, ow), dtype=np.float64)
    for a in range(od):
        for i in range(oh):
            for j in range(ow):
                patch = xp[:, :, a * stride:a * stride + kd, i * stride:i * stride + kh, j * stride:j * stride + kw]
                y[:, :, a, i, j] = np.tensordot(patch, w.astype(np.float64), axes=([1, 2, 3, 4], [1, 2, 3, 4]))
    return y


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-5):
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return (x - mean.reshape(shp)) / np.sqrt(var.reshape(shp) + eps) * gamma.reshape(shp) + beta.reshape(shp)


def maxpool2d(x, k=3, stride=2, pad=1):
    """nn.MaxPool2d: padding behaves as -inf."""
    N, C, H, W = x.shape
    xp = np.full((N, C, H + 2 * pad, W + 2 * pad), -np.inf)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.zeros((N, C, oh, ow))
    for i in range(oh):
        for j in range(ow):
            y[:, :, i, j] = xp[:, :, i * stride:i * stride + k, j * stride:j * stride + k].max((2, 3))
    return y


def bilinear_resize(x, size):
    """F.interpolate(mode='bilinear', align_corners=False, antialias=False) on [N,C,H,W]:
    src = (dst + 0.5) * (in/out) - 0.5 clamped at 0; neighbours clamped at the far edge."""
    N, C, H, W = x.shape
    oh, ow = size
    y = np.zeros((N, C, oh, ow), dtype=np.float32)
    sy, sx = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    for i in range(oh):
        fy = max(np.float32(sy * np.float32(i + 0.5) - np.float32(0.5)), np.float32(0))
        y0 = min(int(fy), H - 1)
        y1 = y0 + (1 if y0 < H - 1 else 0)
        ly = np.float32(min(max(fy - y0, 0), 1))
        for j in range(ow):
            fx = max(np.float32(sx * np.float32(j + 0.5) - np.float32(0.5)), np.float32(0))
            x0 = min(int(fx), W - 1)
            x1 = x0 + (1 if x0 < W - 1 else 0)
            lx = np.float32(min(max(fx - x0, 0), 1))
            top = (1 - lx) * x[:, :, y0, x0] + lx * x[:, :, y0, x1]
            bot = (1 - lx) * x[:, :, y1, x0] + lx * x[:, :, y1, x1]
            y[:, :, i, j] = (1 - ly) * top + ly * bot
    return y


def sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def lstm_bidir(x, params, hidden):
    """One bidirectional nn.LSTM layer (batch_first): x [B,T,I]; params maps
    weight_ih/_hh, bias_ih/_hh (+ '_reverse') -> arrays; gate order i,f,g,o; zero initial state;
    the reverse direction's output at time t sits at index t."""
    B, T, _ = x.shape
    out = np.zeros((B, T, 2 * hidden))
    for d, sfx in enumerate(("", "_reverse")):
        wih, whh = params["weight_ih" + sfx], params["weight_hh" + sfx]
        b = params["bias_ih" + sfx] + params["bias_hh" + sfx]
        h = np.zeros((B, hidden))
        c = np.zeros((B, hidden))
        steps = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in steps:
            g = x[:, t] @ wih.T + h @ whh.T + b
            i, f, gg, o = (g[:, k * hidden:(k + 1) * hidden] for k in range(4))
            c = sigmoid(f) * c + sigmoid(i) * np.tanh(gg)
            h = sigmoid(o) * np.tanh(c)
            out[:, t, d * hidden:(d + 1) * hidden] = h
    return out


def layernorm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def mha_self(x, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention self-attention (batch_first), in_proj rows ordered q,k,v."""
    B, T, D = x.shape
    hd = D // heads
    qkv = x @ in_w.T + in_b
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, T, heads, hd).transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(hd)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
    return o @ out_w.T + out_b


def encoder_layer_postln(x, p, heads):
    """nn.TransformerEncoderLayer(norm_first=False, activation=relu, dropout=0)."""
    a = mha_self(x, p["in_w"], p["in_b"], p["out_w"], p["out_b"], heads)
    x = layernorm(x + a, p["n1_w"], p["n1_b"])
    f = np.maximum(x @ p["l1_w"].T + p["l1_b"], 0) @ p["l2_w"].T + p["l2_b"]
    return layernorm(x + f, p["n2_w"], p["n2_b"])


def tiny_forward(clips, sd):
    """The 'tiny' config end to end: clips [B,T,3,S,S], sd = name -> ndarray."""
    B, T = clips.shape[:2]
    x = clips.reshape((B * T,) + clips.shape[2:]).astype(np.float64)
    x = np.maximum(conv2d(x, sd["backbone.conv1.weight"], sd["backbone.conv1.bias"], 2, 1), 0)
    x = np.maximum(conv2d(x, sd["backbone.conv2.weight"], sd["backbone.conv2.bias"], 2, 1), 0)
    f = x.mean((2, 3)).reshape(B, T, -1).mean(1)
    return f @ sd["fc.weight"].T.astype(np.float64) + sd["fc.bias"]
