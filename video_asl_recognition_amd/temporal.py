"""Temporal encoders over the per-frame feature stack [B,T,F]: 2-layer BiLSTM and the two-stream
(RGB features + hand-keypoint MLP) 4-layer transformer.  Every op is a launch into libaslhip.so;
torch only allocates the buffers."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L
from .packing import PackedConv, pack_linear
from .spec import ModelSpec


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _linear(x, ldx, pc: PackedConv, m, y, ldy, stream, relu=False, out_f32=False, res=None, ldres=0, res_f32=False,
            k=None):
    p = L.LinearParams(m=m, n=pc.cout, k=k or pc.cin_phys, ldx=ldx, ldy=ldy, ldres=ldres, relu=int(relu),
                       out_f32=int(out_f32), res_f32=int(res_f32), w_rows=pc.w_rows)
    L.check(L.lib.asl_linear_bf16(C.byref(p), x, _ptr(pc.w), None, _ptr(pc.bias), res, y, stream), "linear")


class BiLstmHead:
    """nn.LSTM(F, H, num_layers, bidirectional, batch_first) -> mean over T -> Linear(2H, C).

    The gates are sensitive to their pre-activations, so every input-projection GEMM runs on a
    split-precision operand: x as bf16 hi | lo halves against the weight matrix repeated ([W | W]),
    which recovers ~16 mantissa bits of x at twice the (tiny) GEMM cost."""

    def __init__(self, spec: ModelSpec, sd, dev):
        self.spec, self.dev = spec, dev
        H = spec.lstm_hidden
        self.layers = []
        for l in range(spec.lstm_layers):
            w_ih = torch.cat([sd[f"lstm.weight_ih_l{l}"], sd[f"lstm.weight_ih_l{l}_reverse"]], 0).float()
            b = torch.cat([sd[f"lstm.bias_ih_l{l}"] + sd[f"lstm.bias_hh_l{l}"],
                           sd[f"lstm.bias_ih_l{l}_reverse"] + sd[f"lstm.bias_hh_l{l}_reverse"]], 0).float()
            w_hh = torch.stack([sd[f"lstm.weight_hh_l{l}"], sd[f"lstm.weight_hh_l{l}_reverse"]], 0)
            self.layers.append((pack_linear(torch.cat([w_ih, w_ih], 1), dev, bias=b),
                                w_hh.to(torch.bfloat16).contiguous().to(dev)))
        self.fc_w = sd["fc.weight"].float().contiguous().to(dev)
        self.fc_b = sd["fc.bias"].float().contiguous().to(dev)

    def __call__(self, feats, feats_b, ld_b, B, T, keypoints, stream):
        spec, dev, H = self.spec, self.dev, self.spec.lstm_hidden
        x, ldx = _ptr(feats_b), ld_b
        keep = []
        ysum = torch.empty(B, 2 * H, dtype=torch.float32, device=dev)
        state = torch.empty(6 * B * H, dtype=torch.float32, device=dev)
        for li, (pc, w_hh) in enumerate(self.layers):
            gx = torch.empty(B * T, 8 * H, dtype=torch.float32, device=dev)
            _linear(x, ldx, pc, B * T, _ptr(gx), 8 * H, stream, out_f32=True)
            y = torch.empty(B * T, 4 * H, dtype=torch.bfloat16, device=dev)
            last = li == len(self.layers) - 1
            p = L.LstmParams(batch=B, t=T, hidden=H, ld_gx=8 * H, ld_y=4 * H, split_y=1, ysum_scale=1.0 / T)
            L.check(L.lib.asl_lstm_bidir_layer(C.byref(p), _ptr(gx), _ptr(w_hh), _ptr(y), _ptr(ysum) if last else None,
                                               _ptr(state), stream), "lstm")
            keep += [gx, y]
            x, ldx = _ptr(y), 4 * H
        logits = torch.empty(B, spec.num_classes, dtype=torch.float32, device=dev)
        L.check(L.lib.asl_head_f32(_ptr(ysum), B, 2 * H, _ptr(self.fc_w), _ptr(self.fc_b), spec.num_classes,
                                   _ptr(logits), stream), "head")
        self._keep = keep       # the launches above are asynchronous: keep their buffers until the next call
        return logits


def sinusoid_posenc(T: int, d: int) -> torch.Tensor:
    pos = torch.arange(T, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(T, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class TwoStreamTransformerHead:
    """cat(RGB feat, kp_mlp(keypoints)) -> Linear -> + sinusoidal pos-enc -> n_layers post-LN encoder
    layers (fused attention) -> mean over T -> Linear(d, C).  The residual stream is fp32; every GEMM
    operand is its bf16 copy."""

    def __init__(self, spec: ModelSpec, sd, dev):
        self.spec, self.dev = spec, dev
        f = lambda k: sd[k].float()
        self.kp1 = pack_linear(f("kp_mlp.0.weight"), dev, bias=f("kp_mlp.0.bias"), k_phys=64)
        self.kp2 = pack_linear(f("kp_mlp.2.weight"), dev, bias=f("kp_mlp.2.bias"))
        self.fuse = pack_linear(f("fuse.weight"), dev, bias=f("fuse.bias"))
        self.layers = []
        for l in range(spec.n_layers):
            p = f"encoder.layers.{l}"
            self.layers.append(dict(
                qkv=pack_linear(f(p + ".self_attn.in_proj_weight"), dev, bias=f(p + ".self_attn.in_proj_bias")),
                out=pack_linear(f(p + ".self_attn.out_proj.weight"), dev, bias=f(p + ".self_attn.out_proj.bias")),
                l1=pack_linear(f(p + ".linear1.weight"), dev, bias=f(p + ".linear1.bias")),
                l2=pack_linear(f(p + ".linear2.weight"), dev, bias=f(p + ".linear2.bias")),
                n1=(f(p + ".norm1.weight").contiguous().to(dev), f(p + ".norm1.bias").contiguous().to(dev)),
                n2=(f(p + ".norm2.weight").contiguous().to(dev), f(p + ".norm2.bias").contiguous().to(dev))))
        self.fc_w = f("fc.weight").contiguous().to(dev)
        self.fc_b = f("fc.bias").contiguous().to(dev)
        self._pe = {}

    def __call__(self, feats, feats_b, ld_b, B, T, keypoints, stream):
        spec, dev = self.spec, self.dev
        d, M = spec.d_model, B * T
        if keypoints is None or tuple(keypoints.shape[:2]) != (B, T):
            raise ValueError("keypoints [B,T,21,2] required")
        if T > 128:
            raise ValueError("attention kernel handles T <= 128")
        kp = keypoints.to(dev).float().reshape(M, -1).contiguous()
        if kp.shape[1] != spec.kp_dim:
            raise ValueError(f"keypoints flatten to {kp.shape[1]} values per frame, spec wants {spec.kp_dim}")
        if T not in self._pe:
            self._pe[T] = sinusoid_posenc(T, d).to(dev)
        pe = self._pe[T]
        bf, f32 = torch.bfloat16, torch.float32
        new = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)
        kp_b, k1 = new(M, 64), new(M, spec.kp_hidden)
        L.check(L.lib.asl_cast_pad_bf16(_ptr(kp), M, spec.kp_dim, _ptr(kp_b), 64, stream), "cast_pad")
        _linear(_ptr(kp_b), 64, self.kp1, M, _ptr(k1), spec.kp_hidden, stream, relu=True)
        # second MLP layer writes straight into the concat buffer next to the RGB features
        F = spec.feat_dim
        _linear(_ptr(k1), spec.kp_hidden, self.kp2, M, C.c_void_p(feats_b.data_ptr() + 2 * F), ld_b, stream)
        x0 = new(M, d, dt=f32)
        _linear(_ptr(feats_b), ld_b, self.fuse, M, _ptr(x0), d, stream, out_f32=True)
        xf, xb = new(M, d, dt=f32), new(M, d)
        L.check(L.lib.asl_add_rows_f32(_ptr(x0), M, d, _ptr(pe), T, _ptr(xf), _ptr(xb), d, stream), "posenc")
        qkv, att, r, hbuf = new(M, 3 * d), new(M, d), new(M, d, dt=f32), new(M, spec.ffn_dim)
        ap = L.AttnParams(batch=B, t=T, heads=spec.n_heads, head_dim=d // spec.n_heads, ld_qkv=3 * d, ld_out=d)
        for lw in self.layers:
            _linear(_ptr(xb), d, lw["qkv"], M, _ptr(qkv), 3 * d, stream)
            L.check(L.lib.asl_attention_fwd(C.byref(ap), _ptr(qkv), _ptr(att), stream), "attention")
            _linear(_ptr(att), d, lw["out"], M, _ptr(r), d, stream, out_f32=True, res=_ptr(xf), ldres=d, res_f32=True)
            L.check(L.lib.asl_layernorm_f32(_ptr(r), M, d, _ptr(lw["n1"][0]), _ptr(lw["n1"][1]), 1e-5, _ptr(xb), d,
                                            _ptr(xf), stream), "layernorm")
            _linear(_ptr(xb), d, lw["l1"], M, _ptr(hbuf), spec.ffn_dim, stream, relu=True)
            _linear(_ptr(hbuf), spec.ffn_dim, lw["l2"], M, _ptr(r), d, stream, out_f32=True, res=_ptr(xf), ldres=d,
                    res_f32=True)
            L.check(L.lib.asl_layernorm_f32(_ptr(r), M, d, _ptr(lw["n2"][0]), _ptr(lw["n2"][1]), 1e-5, _ptr(xb), d,
                                            _ptr(xf), stream), "layernorm")
        pooled = new(B, d, dt=f32)
        L.check(L.lib.asl_mean_t_f32(_ptr(xf), B, T, d, _ptr(pooled), stream), "mean_t")
        logits = new(B, spec.num_classes, dt=f32)
        L.check(L.lib.asl_head_f32(_ptr(pooled), B, d, _ptr(self.fc_w), _ptr(self.fc_b), spec.num_classes,
                                   _ptr(logits), stream), "head")
        self._keep = [kp, kp_b, k1, x0, xf, xb, qkv, att, r, hbuf, pooled]
        return logits


def build_temporal(spec: ModelSpec, sd, dev):
    if spec.temporal == "bilstm":
        return BiLstmHead(spec, sd, dev)
    if spec.temporal == "transformer":
        return TwoStreamTransformerHead(spec, sd, dev)
    raise NotImplementedError(spec.temporal)
