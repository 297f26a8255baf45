"""I3D-style 3-D conv backbone on the gather-GEMM: every Unit3D (Conv3d + BatchNorm3d + ReLU) is one
launch over a halo'd channels-last (NDHWC) bf16 tensor, Inception branches write their channel slice
of the block's concatenated output directly, and 3-D max-pools read the zero halo as padding."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List

import torch

from . import _lib as L
from .packing import PackedConv, bn_scale_bias, pack_conv
from .spec import I3D_MIXED, ModelSpec


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _r32(c):
    return (c + 31) // 32 * 32


class Act3:
    """[n][d+2*hd][h+2*hh][w+2*hh][c] bf16, zero halo; c is the PHYSICAL channel count (multiple of 32,
    channels beyond the logical count stay zero)."""

    def __init__(self, n, d, h, w, c, hd, hh, dev):
        self.n, self.d, self.h, self.w, self.c, self.hd, self.hh = n, d, h, w, c, hd, hh
        self.t = torch.zeros(n, d + 2 * hd, h + 2 * hh, w + 2 * hh, c, dtype=torch.bfloat16, device=dev)

    @property
    def key(self):
        return (self.d, self.h, self.w, self.c, self.hd, self.hh)


class _Pool3:
    def __init__(self, n, dev):
        self.n, self.dev, self.free, self.all = n, dev, {}, []

    def get(self, d, h, w, c, hd=1, hh=1) -> Act3:
        lst = self.free.get((d, h, w, c, hd, hh))
        if lst:
            return lst.pop()
        a = Act3(self.n, d, h, w, c, hd, hh, self.dev)
        self.all.append(a)
        return a

    def put(self, a):
        self.free.setdefault(a.key, []).append(a)


def pack_i3d_weights(spec: ModelSpec, sd, dev) -> Dict[str, PackedConv]:
    w: Dict[str, PackedConv] = {}

    def unit(name, cin_phys=None):
        s, b = bn_scale_bias(sd, f"backbone.{name}.bn", 1e-3, dev)
        w[name] = pack_conv(sd[f"backbone.{name}.conv.weight"], dev, cin_phys=cin_phys, scale=s, bias=b)

    # stem: [64,3,7,7,7] -> taps (kd, kh) x (8 pixels x 4 channels)
    ws = sd["backbone.conv_1a.conv.weight"].float()
    cout = ws.shape[0]
    p = torch.zeros((cout + 127) // 128 * 128, 7, 7, 8, 4)
    p[:cout, :, :, :7, :3] = ws.permute(0, 2, 3, 4, 1)
    s, b = bn_scale_bias(sd, "backbone.conv_1a.bn", 1e-3, dev)
    w["conv_1a"] = PackedConv(p.reshape(p.shape[0], 49 * 32).to(torch.bfloat16).contiguous().to(dev), s, b, cout, 4,
                              (7, 7, 7), p.shape[0])
    unit("conv_2b")
    unit("conv_2c")
    # conv_2c (64 -> 192 at 56 x 56) runs on the gather-GEMM, whose channel tiles are 128 or 64 wide: as one launch its
    # second tile multiplies 64 rows of zero padding; as a 128-wide and a 64-wide launch into the two channel slices of
    # the output it does not (ASL_I3D_SPLIT_2C=0 keeps the single launch)
    s2c, b2c = bn_scale_bias(sd, "backbone.conv_2c.bn", 1e-3, dev)
    w2c = sd["backbone.conv_2c.conv.weight"]
    for tag, lo, hi in (("lo", 0, 128), ("hi", 128, 192)):
        w["conv_2c." + tag] = pack_conv(w2c[lo:hi], dev, scale=s2c[lo:hi].contiguous(), bias=b2c[lo:hi].contiguous())
    cin = 192
    for name, wd in I3D_MIXED:
        if wd is None:
            continue
        cp = _r32(cin)
        unit(name + ".b0", cp)
        unit(name + ".b1a", cp)
        unit(name + ".b1b", _r32(wd[1]))
        unit(name + ".b2a", cp)
        unit(name + ".b2b", _r32(wd[3]))
        unit(name + ".b3b", cp)
        cin = wd[0] + wd[2] + wd[4] + wd[5]
    return w


class I3dPlan:
    """Launch plan for up to `nmax` clips of T frames at SxS."""

    def __init__(self, spec: ModelSpec, w: Dict[str, PackedConv], nmax: int, T: int, dev):
        self.spec, self.nmax, self.T, self.dev = spec, nmax, T, dev
        S = spec.size
        pool = _Pool3(nmax, dev)
        self.ops: List = []            # every launch, in single-stream order: op(n, stream)
        # The same launches as a schedule over three streams: ("op", index into ops, stream) | ("fork",) | ("join",).
        # The branches of an Inception block are independent until the concatenated output is complete, and from the
        # 14 x 14 layers on a single launch no longer fills the chip, so branch 1 runs on stream 0, branches 2 and 0 on
        # stream 1 and the pooled branch on stream 2; "fork" makes streams 1 and 2 wait for stream 0, "join" the reverse.
        self.sched: List = []
        self.x0 = torch.zeros(nmax, T + 6, S + 6, S + 6, 4, dtype=torch.bfloat16, device=dev)

        def conv(pc: PackedConv, x: Act3, y: Act3, coff=0, stride=1):
            kd, kh, kw = pc.k
            p = L.Conv3dParams(n=0, d=x.d, h=x.h, w=x.w, cin=x.c, in_halo_d=x.hd, in_halo=x.hh, cout=pc.cout,
                               kd=kd, kh=kh, kw=kw, stride_d=stride, stride=stride, pad_d=kd // 2, pad=kh // 2,
                               out_halo_d=y.hd, out_halo=y.hh, out_ctot=y.c, out_coff=coff, relu=1, w_rows=pc.w_rows)
            args = (_ptr(x.t), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.bias), _ptr(y.t))

            def run(n, stream, p=p, args=args, keep=(x, y, pc)):
                p.n = n
                L.check(L.lib.asl_conv3d_ndhwc_bf16(C.byref(p), *args, stream), "conv3d")
            self.ops.append(run)
            return len(self.ops) - 1

        def maxpool(x: Act3, y: Act3, k, s, pad):
            p = L.Pool3dParams(n=0, d=x.d, h=x.h, w=x.w, c=x.c, in_halo_d=x.hd, in_halo=x.hh, kd=k[0], kh=k[1], kw=k[2],
                               sd=s[0], sh=s[1], sw=s[2], pd=pad[0], ph=pad[1], pw=pad[2], out_halo_d=y.hd,
                               out_halo=y.hh)

            def run(n, stream, p=p, keep=(x, y)):
                p.n = n
                L.check(L.lib.asl_maxpool3d_ndhwc(C.byref(p), _ptr(x.t), _ptr(y.t), stream), "maxpool3d")
            self.ops.append(run)
            return len(self.ops) - 1

        # ---- stem 7x7x7 / 2 on packed RGB: a gather-GEMM with one tap per (kd, kh) filter row ----------
        D1, H1 = (T + 6 - 7) // 2 + 1, S // 2
        a1 = pool.get(D1, H1, H1, 64, 0, 1)
        pc = w["conv_1a"]
        wp, hp = S + 6, S + 6
        plane = hp * wp * 4
        g = L.GemmParams()
        g.nb, g.od, g.oh, g.ow, g.cout = 0, D1, H1, H1, 64
        g.kd, g.kh, g.kw, g.ck = 7, 7, 1, 32
        g.tap_sd, g.tap_sh, g.tap_sw = plane, wp * 4, 0
        g.in_.sn, g.in_.sd, g.in_.sh, g.in_.sw, g.in_.off0 = plane * (T + 6), 2 * plane, 2 * wp * 4, 8, 0
        owp = H1 + 2
        oplane = owp * owp * 64
        g.out.sn, g.out.sd, g.out.sh, g.out.sw, g.out.off0 = oplane * D1, oplane, owp * 64, 64, (owp + 1) * 64
        g.relu, g.out_f32, g.res_f32, g.w_rows = 1, 0, 0, pc.w_rows
        sargs = (_ptr(self.x0), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.bias), None, _ptr(a1.t))

        # the LDS-resident-patch stem kernel takes the same tensors; the gather-GEMM form above is the fallback for sizes
        # whose tiles do not fit it, and the ASL_I3D_GATHER_STEM=1 A/B switch
        sp = L.Stem3dParams(n=0, frames=T, size=S, cout=64, out_halo_d=a1.hd, out_halo=a1.hh, relu=1, w_rows=pc.w_rows)
        s3args = (_ptr(self.x0), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.bias), _ptr(a1.t))
        use_gather = [os.environ.get("ASL_I3D_GATHER_STEM", "0") == "1" or T % 2 == 1]

        def stem(n, stream, g=g, sargs=sargs, keep=(a1, pc)):
            if not use_gather[0]:
                sp.n = n
                rc = L.lib.asl_stem3d_bf16(C.byref(sp), *s3args, stream)
                if rc != L.ASL_ENOTSUP:
                    L.check(rc, "i3d stem")
                    return
                use_gather[0] = True
            g.nb = n
            g.x_elems = n * g.in_.sn
            g.y_elems = n * g.out.sn
            L.check(L.lib.asl_gather_gemm_bf16(C.byref(g), *sargs, stream), "i3d stem")
        self.ops.append(stem)

        a2 = pool.get(D1, H1 // 2, H1 // 2, 64, 0, 0)
        maxpool(a1, a2, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        pool.put(a1)
        a3 = pool.get(a2.d, a2.h, a2.w, 64, 1, 1)
        conv(w["conv_2b"], a2, a3)
        pool.put(a2)
        a4 = pool.get(a3.d, a3.h, a3.w, 192, 0, 1)
        if os.environ.get("ASL_I3D_SPLIT_2C", "1") == "1" and w["conv_2c"].cout == 192:
            conv(w["conv_2c.lo"], a3, a4, 0)
            conv(w["conv_2c.hi"], a3, a4, 128)
        else:
            conv(w["conv_2c"], a3, a4)
        pool.put(a3)
        x = pool.get(a4.d, a4.h // 2, a4.w // 2, 192, 1, 1)
        maxpool(a4, x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        pool.put(a4)

        for name, wd in I3D_MIXED:
            if name == "pool_4a":
                y = pool.get((x.d + 2 - 3) // 2 + 1, (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1, x.c, 1, 1)
                maxpool(x, y, (3, 3, 3), (2, 2, 2), (1, 1, 1))
            elif name == "pool_5a":
                if x.d < 2:
                    raise ValueError("clip too short for the I3D pooling pyramid (needs T >= 16)")
                y = pool.get(x.d // 2, x.h // 2, x.w // 2, x.c, 1, 1)
                maxpool(x, y, (2, 2, 2), (2, 2, 2), (0, 0, 0))
            else:
                ctot = wd[0] + wd[2] + wd[4] + wd[5]
                halo = 0 if name == "mixed_5c" else 1          # the last block feeds the global average pool
                y = pool.get(x.d, x.h, x.w, _r32(ctot), halo, halo)
                first = len(self.ops)
                self.sched += [("op", i, 0) for i in range(len(self.sched_done()), first)]
                b0 = conv(w[name + ".b0"], x, y, 0)
                t1 = pool.get(x.d, x.h, x.w, _r32(wd[1]), 1, 1)
                b1a = conv(w[name + ".b1a"], x, t1)
                b1b = conv(w[name + ".b1b"], t1, y, wd[0])
                t2 = pool.get(x.d, x.h, x.w, _r32(wd[3]), 1, 1)
                b2a = conv(w[name + ".b2a"], x, t2)
                b2b = conv(w[name + ".b2b"], t2, y, wd[0] + wd[2])
                t3 = pool.get(x.d, x.h, x.w, x.c, 0, 0)
                b3a = maxpool(x, t3, (3, 3, 3), (1, 1, 1), (1, 1, 1))
                b3b = conv(w[name + ".b3b"], t3, y, wd[0] + wd[2] + wd[4])
                self.sched += [("fork",), ("op", b1a, 0), ("op", b2a, 1), ("op", b3a, 2), ("op", b1b, 0), ("op", b2b, 1),
                               ("op", b3b, 2), ("op", b0, 1), ("join",)]
                for t in (t1, t2, t3):
                    pool.put(t)
            pool.put(x)
            x = y
        self.sched += [("op", i, 0) for i in range(len(self.sched_done()), len(self.ops))]
        assert sorted(self.sched_done()) == list(range(len(self.ops)))
        self.last = x
        self.pool = pool

    def sched_done(self):
        return [e[1] for e in self.sched if e[0] == "op"]

    def nbytes(self):
        return sum(a.t.numel() * 2 for a in self.pool.all) + self.x0.numel() * 2
