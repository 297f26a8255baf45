"""AslClipModel: the clip -> logits path on one MI355X, driven from Python.

Surface (the "forward()/classify-clip API" the path must keep):
    model = AslClipModel(spec)                 # spec: ModelSpec or its name
    model.load_state_dict(sd)                  # oracle / torchvision-style parameter names
    logits = model(clips, keypoints=None)      # clips [B,T,3,S,S] float -> [B,C] float32
    idx, prob = model.classify_clip(frames_u8, keypoints=None, topk=5)

PyTorch supplies device memory and the stream; every op on the path is a launch into
libaslhip.so through ``_lib``.  Frames run through the backbone in chunks of
``frames_per_chunk`` so activation buffers (channels-last bf16 with a zero halo, allocated
once and reused) stay a fixed size whatever B*T is.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .packing import (pack_dual1x1, PackedConv, bn_scale_bias, pack_conv, pack_entry_conv, pack_entry_ds, pack_linear, pack_stem,
                      pack_stem_planar)
from .spec import RESNET_LAYERS, SPECS, ModelSpec, param_shapes, resnet_blocks

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Act:
    """Channels-last bf16 activation buffer with a zero halo: [n][h+2*halo][w+2*halo][c]; with cells=True the same
    halo-1 image as space-to-depth cells [n][(h+2)/2][(w+2)/2][4][c] (the input layout of a stride-2 block entry)."""

    def __init__(self, n, h, w, c, halo, device, cells=False):
        self.n, self.h, self.w, self.c, self.halo, self.cells = n, h, w, c, halo, cells
        if cells:
            assert halo == 1 and h % 2 == 0 and w % 2 == 0
            self.t = torch.zeros(n, (h + 2) // 2, (w + 2) // 2, 4, c, dtype=torch.bfloat16, device=device)
        else:
            self.t = torch.zeros(n, h + 2 * halo, w + 2 * halo, c, dtype=torch.bfloat16, device=device)

    @property
    def key(self):
        return (self.h, self.w, self.c, self.halo, self.cells)

    def interior(self, n=None):
        hl = self.halo
        t = self.t[: (n or self.n)]
        return t[:, hl:hl + self.h, hl:hl + self.w] if hl else t


class _BufPool:
    def __init__(self, n, device):
        self.n, self.device = n, device
        self.free: Dict[tuple, List[Act]] = {}
        self.all: List[Act] = []

    def get(self, h, w, c, halo=1, cells=False) -> Act:
        lst = self.free.get((h, w, c, halo, cells))
        if lst:
            return lst.pop()
        a = Act(self.n, h, w, c, halo, self.device, cells)
        self.all.append(a)
        return a

    def put(self, a: Act):
        self.free.setdefault(a.key, []).append(a)

    def nbytes(self):
        return sum(a.t.numel() * 2 for a in self.all)


class AslClipModel:
    def __init__(self, spec, device: str = "cuda", frames_per_chunk: Optional[int] = None, seed: Optional[int] = None,
                 streams: int = 2, s2d_entry: Optional[bool] = None, parity_mode: Optional[bool] = None,
                 chunk_heads: Optional[bool] = None):
        self.spec: ModelSpec = SPECS[spec] if isinstance(spec, str) else spec
        if not torch.cuda.is_available():
            raise RuntimeError("AslClipModel needs an MI355X (no GPU visible); there is no CPU path")
        self.device = torch.device(device)
        # frames per backbone launch.  Default 256 for the per-frame backbones; 1 024 for I3D (16 clips of 64 frames per
        # launch): its deep layers are a few hundred pixels per clip, and four clips per launch left most CUs without a
        # tile (16 x 64: 10.95 -> 8.01 ms per forward)
        # A batch that fits one chunk is still divided over the streams when each part keeps at least split_frames
        # frames (ResNet-50: 8 x 64 as 2 x 256 on two streams 8.07 ms, as 1 x 512 8.34; ResNet-18: 10 x 64 as 1 x 640
        # 2.45 ms, as 2 x 320 2.47 -- never split)
        self.split_frames = None
        if frames_per_chunk is None:
            # frames per backbone launch: enough that the late, small layers fill whole rounds of workgroups (ResNet-18:
            # 640 frames = 1 960 / 980 / 256 tiles on 512 / 512 / 256 slots; I3D: 16 clips; ResNet-50: 16 x 64 as
            # 2 x 512 15.3 ms, as 4 x 256 15.55, 32 x 64 29.1 against 29.55)
            frames_per_chunk = {"i3d": 1024, "resnet18": 640, "resnet50": 512}.get(self.spec.backbone, 256)
            if self.spec.backbone == "resnet50":
                self.split_frames = 256
        self.frames_per_chunk = int(frames_per_chunk)
        if self.split_frames is None:
            self.split_frames = self.frames_per_chunk
        self.streams = int(streams)
        self._streams: List[torch.cuda.Stream] = []
        self._i3d_events: List[torch.cuda.Event] = []
        self._i3d_plans: List = []
        self._plan = None
        self._groups: Dict[int, tuple] = {}
        # diagnostics / A-B runs: force one conv kernel variant (L.CONV_*), run stem and max-pool unfused
        self.conv_variant = int(os.environ.get("ASL_CONV_VARIANT", "0"))
        self.fused_stem = not os.environ.get("ASL_NO_FUSED_STEM")
        # one kernel per 64-channel BasicBlock (asl_basic_block_c64_bf16) instead of two conv launches: half the HBM
        # traffic of layer 1; ASL_FUSE_BLOCKS=0 keeps the two-launch form
        self.fuse_blocks = os.environ.get("ASL_FUSE_BLOCKS", "1") == "1"
        # stride-2 BasicBlocks: conv1 and the downsample from one launch over a cell copy of the block's input that the
        # previous block's last kernel writes (asl_block_entry_s2_bf16).  Opt-in (argument, or ASL_S2D_ENTRY=1): measured
        # slower than the two gather-GEMM launches it replaces (254 / 188 / 145 us against 190 / 122 / 105 at 640 frames)
        self.s2d_entry = (os.environ.get("ASL_S2D_ENTRY", "0") == "1") if s2d_entry is None else bool(s2d_entry)
        # Parity mode (argument, or ASL_PARITY_MODE=1; ResNet backbones): every activation travels as a bf16 pair hi + lo
        # (~16 mantissa bits) instead of one bf16 value, every convolution reads both halves against weights duplicated
        # along K, and accumulation, BatchNorm, residual and ReLU stay fp32 as always -- twice the arithmetic and traffic,
        # for inputs whose conditioning makes single-bf16 activations miss the 1e-3 bar.  All convs on the gather-GEMM.
        self.parity_mode = (os.environ.get("ASL_PARITY_MODE", "0") == "1") if parity_mode is None else bool(parity_mode)
        if self.parity_mode and self.spec.backbone not in RESNET_LAYERS:
            raise NotImplementedError("parity mode covers the ResNet backbones")
        # ResNet stem straight from the bf16 clip tensor (no packed, halo'd copy of the input): ASL_PLANAR_STEM=0 goes
        # back to pack + packed-input stem (and uint8 frames are then resized into the packed form instead of into clips)
        self.planar_stem = os.environ.get("ASL_PLANAR_STEM", "1") != "0"
        # ResNet-50, 56 x 56 stage: a bottleneck's last 1x1 conv (+ shortcut + ReLU) and the next bottleneck's first 1x1 conv
        # as one launch; ASL_CHAIN_1X1=0 keeps them apart (same results bit for bit)
        self.chain_1x1 = os.environ.get("ASL_CHAIN_1X1", "1") != "0"
        # BiLSTM / transformer heads per chunk of whole clips, on the chunk's stream, under the other stream's backbone
        # launches instead of one head over the whole batch after the backbone.  Opt-in (argument, or ASL_CHUNK_HEADS=1):
        # measured no faster -- ResNet-50 + BiLSTM 16 x 64 17.65 against 16.39 ms (the recurrence's 128 short launches
        # per chunk sit in front of the stream's next chunk), two-stream transformer 8 x 128 4.23 ms either way
        self.chunk_heads = (os.environ.get("ASL_CHUNK_HEADS", "0") == "1") if chunk_heads is None else bool(chunk_heads)
        # ResNet-50: the last 1x1 conv of a bottleneck and its projected shortcut (1x1, stride 1 or 2) as ONE launch, the
        # shortcut's K steps first and its sum rescaled in the accumulator (asl_conv1x1_dual_bf16): the shortcut tensor is
        # neither written nor read back.  ASL_FUSE_SHORTCUT=0 keeps the two launches.
        self.fuse_shortcut = os.environ.get("ASL_FUSE_SHORTCUT", "1") == "1"
        self._keep_heads = None
        self._w: Dict[str, PackedConv] = {}
        self._fc_w = self._fc_b = None
        self._temporal = None
        if seed is not None:
            self.load_state_dict(init_state_dict(self.spec, seed))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        shapes = param_shapes(self.spec)
        sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
        if strict:
            missing = sorted(set(shapes) - set(sd))
            unexpected = sorted(set(sd) - set(shapes))
            if missing or unexpected:
                raise KeyError(f"state_dict mismatch: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != shp:
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shp}")
        dev, spec, w = self.device, self.spec, {}
        if spec.backbone == "tiny":
            w["stem"] = pack_stem(sd["backbone.conv1.weight"], dev,
                                  bias=sd["backbone.conv1.bias"].float().contiguous().to(dev))
            w["conv2"] = pack_conv(sd["backbone.conv2.weight"], dev,
                                   bias=sd["backbone.conv2.bias"].float().contiguous().to(dev))
        elif spec.backbone in RESNET_LAYERS:
            s, b = bn_scale_bias(sd, "backbone.bn1", 1e-5, dev)
            w["stem"] = pack_stem(sd["backbone.conv1.weight"], dev, s, b)
            w["stem_planar"] = pack_stem_planar(sd["backbone.conv1.weight"], dev, s, b)
            nconv = 2 if RESNET_LAYERS[spec.backbone][0] == "basic" else 3
            split = self.parity_mode
            for prefix, cin, planes, stride, ds in resnet_blocks(spec.backbone):
                for i in range(1, nconv + 1):
                    s, b = bn_scale_bias(sd, f"{prefix}.bn{i}", 1e-5, dev)
                    w[f"{prefix}.conv{i}"] = pack_conv(sd[f"{prefix}.conv{i}.weight"], dev, scale=s, bias=b, split=split)
                if ds:
                    s, b = bn_scale_bias(sd, f"{prefix}.downsample.1", 1e-5, dev)
                    w[f"{prefix}.ds"] = pack_conv(sd[f"{prefix}.downsample.0.weight"], dev, scale=s, bias=b, split=split)
                    if not split and nconv == 2 and stride == 2 and cin % 32 == 0 and planes % 128 == 0:
                        # the same two weight sets in MFMA-fragment order for the one-launch block entry
                        c1 = w[f"{prefix}.conv1"]
                        w[f"{prefix}.entry"] = pack_entry_conv(sd[f"{prefix}.conv1.weight"], dev, c1.scale, c1.bias)
                        w[f"{prefix}.entry_ds"] = pack_entry_ds(sd[f"{prefix}.downsample.0.weight"], dev, s, b)
                    if not split and nconv == 3 and self.fuse_shortcut:
                        # bottleneck exit + projected shortcut as one launch (asl_conv1x1_dual_bf16)
                        c3 = w[f"{prefix}.conv3"]
                        dual = pack_dual1x1(sd[f"{prefix}.conv3.weight"], c3.scale, c3.bias,
                                            sd[f"{prefix}.downsample.0.weight"], s, b, dev)
                        if dual is not None:
                            w[f"{prefix}.dual"], w[f"{prefix}.dual_pre"] = dual
        elif spec.backbone == "i3d":
            from .i3d import pack_i3d_weights
            w = pack_i3d_weights(spec, sd, dev)
            self._fc_w = sd["backbone.logits.weight"].float().reshape(spec.num_classes, -1).contiguous().to(dev)
            self._fc_b = sd["backbone.logits.bias"].float().contiguous().to(dev)
        else:
            raise NotImplementedError(f"backbone {spec.backbone}")
        self._w = w
        if spec.temporal == "i3d":
            pass
        elif spec.temporal == "mean":
            self._fc_w = sd["fc.weight"].float().contiguous().to(dev)
            self._fc_b = sd["fc.bias"].float().contiguous().to(dev)
        else:
            from .temporal import build_temporal
            self._temporal = build_temporal(spec, sd, dev)
        self._plan = None
        self._i3d_plans = []
        self._groups = {}
        return self

    # ------------------------------------------------------------------ plan
    def _build_plan(self, nmax: int):
        """The backbone of one chunk of up to `nmax` frames as data: a list of asl_op records (L.Op) over
        buffers allocated here, issued by one asl_run_ops call per group of chunks."""
        spec, dev = self.spec, self.device
        pool = _BufPool(nmax, dev)
        ops = []            # dicts: op (L.Op template), params (struct with a field n, or None), n_slot (index of n
        S = spec.size       # in op.i, or None), conv (priced by the roofline), keep (tensors the pointers refer to)

        def add(kind, params, ptrs, ints=(), n_slot=None, conv=False, keep=()):
            ops.append(dict(op=L.make_op(kind, params, [t.data_ptr() if t is not None else None for t in ptrs], ints),
                            params=params, n_slot=n_slot, conv=conv, keep=(params, ptrs, keep)))

        pm = self.parity_mode     # parity mode: Acts hold [hi | lo] halves (2 x the logical channels), gather-GEMM only

        def conv_op(pc: PackedConv, x: Act, y: Act, stride, pad, relu, res: Optional[Act] = None):
            kd, kh, kw = pc.k
            p = L.Conv2dParams(n=0, h=x.h, w=x.w, cin=x.c, in_halo=x.halo, cout=pc.cout, kh=kh, kw=kw,
                               stride=stride, pad=pad, out_halo=y.halo, out_ctot=y.c, out_coff=0,
                               res_halo=res.halo if res else 0, res_ctot=res.c if res else 0, res_coff=0,
                               relu=int(relu), w_rows=pc.w_rows, variant=L.CONV_GATHER if pm else self.conv_variant,
                               out_s2d=int(y.cells), out_split=int(pm))
            add(L.OP_CONV2D, p, (x.t, pc.w, pc.scale, pc.bias, res.t if res else None, y.t), conv=True, keep=(x, y, res, pc))

        stem_planar = None

        def stem_op(pc: PackedConv, x0: torch.Tensor, in_halo, y: Act, stride, pad, relu):
            kd, kh, kw = pc.k
            p = L.StemParams(n=0, h=S, w=S, in_halo=in_halo, cout=pc.cout, kh=kh, kw=kw, stride=stride, pad=pad,
                             out_halo=y.halo, relu=int(relu), w_rows=pc.w_rows, out_split=int(pm))
            add(L.OP_STEM_CONV, p, (x0, pc.w, pc.scale, pc.bias, y.t), conv=True, keep=(y, pc))

        if spec.backbone == "tiny":
            in_halo = 5         # (ow-1)*2 - 1 + 8 - S for the 3x3/2/1 first layer
            x0 = torch.zeros(nmax, S + 2 * in_halo, S + 2 * in_halo, 4, dtype=torch.bfloat16, device=dev)
            a = pool.get(S // 2, S // 2, 32)
            stem_op(self._w["stem"], x0, in_halo, a, 2, 1, True)
            x = pool.get(S // 4, S // 4, 64)
            conv_op(self._w["conv2"], a, x, 2, 1, True)
        else:
            in_halo = 3
            x0 = torch.zeros(nmax, S + 6, S + 6, 4, dtype=torch.bfloat16, device=dev)
            cm = 2 if pm else 1                               # channel multiple of every activation buffer
            x = pool.get(S // 4, S // 4, 64 * cm)
            if pm:
                a = pool.get(S // 2, S // 2, 128)
                stem_op(self._w["stem"], x0, in_halo, a, 2, 3, True)
                add(L.OP_MAXPOOL_SPLIT, None, (a.t, x.t), (0, a.h, a.w, 64, a.halo, x.halo), n_slot=0, keep=(a, x))
                pool.put(a)
            elif not self.fused_stem:
                a = pool.get(S // 2, S // 2, 64)
                stem_op(self._w["stem"], x0, in_halo, a, 2, 3, True)
                add(L.OP_MAXPOOL, None, (a.t, x.t), (0, a.h, a.w, a.c, a.halo, x.halo), n_slot=0, keep=(a, x))
                pool.put(a)
            else:
                pc = self._w["stem"]
                sp = L.StemPoolParams(n=0, h=S, w=S, cout=pc.cout, out_halo=x.halo, w_rows=pc.w_rows)
                add(L.OP_STEM_POOL, sp, (x0, pc.w, pc.scale, pc.bias, x.t), conv=True, keep=(x, pc))
                # the same launch reading the bf16 clip tensor itself (no packed copy): swapped in per call when the
                # caller hands over clips rather than uint8 frames; its source pointer is set per chunk
                pp = self._w["stem_planar"]
                spp = L.StemPoolParams(n=0, h=S, w=S, cout=pp.cout, out_halo=x.halo, w_rows=pp.w_rows, planar=1)
                stem_planar = dict(op=L.make_op(L.OP_STEM_POOL, spp, [None, pp.w.data_ptr(), pp.scale.data_ptr(),
                                                                     pp.bias.data_ptr(), x.t.data_ptr()]),
                                   params=spp, index=len(ops) - 1, keep=(pp, x))
            basic = RESNET_LAYERS[spec.backbone][0] == "basic"
            blocks = list(resnet_blocks(spec.backbone))
            cells_ok = basic and self.s2d_entry and self.conv_variant in (0, L.CONV_PP) and not pm
            t1_pre = None                                         # a bottleneck's conv1 output, when the block before wrote it
            for bi, (prefix, cin, planes, stride, ds) in enumerate(blocks):
                oh, ow = x.h // stride, x.w // stride
                w = self._w
                idt = x                                           # the block's shortcut: its input, or a projection of it
                # the next block is a stride-2 entry that can take its input as cells, and this block's last kernel can
                # write them (the paired fused block; the 8-wave 3x3 kernel: 128 channels and up, whole 128-channel tiles)
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
                fused = basic and self.fuse_blocks and not ds and stride == 1 and cin == planes == 64 and not pm
                y_cells = bool(cells_ok and nxt and (nxt[0] + ".entry") in w and oh % 2 == 0 and ow % 2 == 0 and
                               (fused or (planes >= 128 and planes % 128 == 0)))
                if fused:
                    # both convs of the block in one kernel: the intermediate activation stays on the chip
                    c1, c2 = w[prefix + ".conv1"], w[prefix + ".conv2"]
                    y = pool.get(oh, ow, planes, cells=y_cells)
                    bp = L.BlockParams(n=0, h=x.h, w=x.w, channels=64, in_halo=x.halo, out_halo=y.halo, w_rows=c1.w_rows,
                                       variant=2 if y_cells else 0, out_s2d=int(y_cells))
                    add(L.OP_BASIC_BLOCK, bp, (x.t, c1.w, c1.scale, c1.bias, c2.w, c2.scale, c2.bias, y.t), conv=True,
                        keep=(x, y, c1, c2))
                elif basic:
                    t = pool.get(oh, ow, planes * cm)
                    idt = x
                    if x.cells:
                        # conv1 (3x3 / stride 2) and the downsample (1x1 / stride 2) from one launch over the cells
                        e1, e2 = w[prefix + ".entry"], w[prefix + ".entry_ds"]
                        idt = pool.get(oh, ow, planes)
                        ep = L.EntryParams(n=0, h=x.h, w=x.w, cin=x.c, cout=planes, out_halo=t.halo, ds_halo=idt.halo, relu=1,
                                           w_rows=planes)
                        add(L.OP_BLOCK_ENTRY, ep, (x.t, e1.w, e1.scale, e1.bias, e2.w, e2.scale, e2.bias, t.t, idt.t),
                            conv=True, keep=(x, t, idt, e1, e2))
                    else:
                        conv_op(w[prefix + ".conv1"], x, t, stride, 1, True)
                        if ds:
                            idt = pool.get(oh, ow, planes * cm)
                            conv_op(w[prefix + ".ds"], x, idt, stride, 0, False)
                    y = pool.get(oh, ow, planes * cm, cells=y_cells)
                    conv_op(w[prefix + ".conv2"], t, y, 1, 1, True, res=idt)
                    pool.put(t)
                else:
                    if t1_pre is not None:                        # written by the previous block's last launch
                        t1, t1_pre = t1_pre, None
                    else:
                        t1 = pool.get(x.h, x.w, planes * cm)
                        conv_op(w[prefix + ".conv1"], x, t1, 1, 0, True)
                    t2 = pool.get(oh, ow, planes * cm)
                    conv_op(w[prefix + ".conv2"], t1, t2, stride, 1, True)
                    idt = x
                    y = pool.get(oh, ow, planes * 4 * cm)
                    if ds and (prefix + ".dual") in w and self.conv_variant in (0, L.CONV_GATHER):
                        # conv3 and the projected shortcut in one launch: the shortcut tensor is never materialised
                        pc, pre = w[prefix + ".dual"], w[prefix + ".dual_pre"]
                        dp = L.Dual1x1Params(n=0, h=oh, w=ow, cin=t2.c, in_halo=t2.halo, cin2=x.c, in2_halo=x.halo, stride2=stride,
                                             cout=pc.cout, out_halo=y.halo, out_ctot=y.c, out_coff=0, relu=1, w_rows=pc.w_rows)
                        add(L.OP_CONV1X1_DUAL, dp, (t2.t, x.t, pc.w, pc.scale, pc.bias, pre, y.t), conv=True, keep=(t2, x, y, pc, pre))
                    elif (self.chain_1x1 and not ds and not pm and nxt is not None and self.conv_variant in (0, L.CONV_GATHER)
                          and t2.c == 64 and y.c == 256 and nxt[2] in (64, 128, 256)):
                        # conv3 + shortcut of this block and conv1 of the next in one launch (asl_conv1x1_chain_bf16): the
                        # 256-channel tensor is written once and not read back for the next block's first conv
                        ca, cb = w[prefix + ".conv3"], w[nxt[0] + ".conv1"]
                        t1_pre = pool.get(oh, ow, nxt[2])
                        sba, sbb = torch.cat([ca.scale, ca.bias]).contiguous(), torch.cat([cb.scale, cb.bias]).contiguous()
                        cp = L.Chain1x1Params(n=0, h=oh, w=ow, cin=t2.c, cmid=y.c, cout=nxt[2], in_halo=t2.halo, res_halo=x.halo,
                                              mid_halo=y.halo, out_halo=t1_pre.halo, w_rows_a=ca.w_rows, w_rows_b=cb.w_rows)
                        add(L.OP_CONV1X1_CHAIN, cp, (t2.t, ca.w, sba, x.t, y.t, cb.w, sbb, t1_pre.t), conv=True,
                            keep=(t2, x, y, t1_pre, ca, cb, sba, sbb))
                    else:
                        if ds:
                            idt = pool.get(oh, ow, planes * 4 * cm)
                            conv_op(w[prefix + ".ds"], x, idt, stride, 0, False)
                        conv_op(w[prefix + ".conv3"], t2, y, 1, 0, True, res=idt)
                    pool.put(t1)
                    pool.put(t2)
                if idt is not x:
                    pool.put(idt)
                pool.put(x)
                x = y
        plan = {"x0": x0, "in_halo": in_halo, "ops": ops, "last": x, "pool": pool, "nmax": nmax, "stem_planar": stem_planar}
        return plan

    def _group_array(self, plans, g: int):
        """asl_op array for a group of g chunks (chunk j on plan j and stream j): g fill ops, then the plans'
        ops interleaved op by op across the group (so every stream gets work early), then g avg-pools.  Built
        once per group size; per call only n, the fill source and the pool destinations change."""
        cache = self._groups
        if g not in cache:
            nops = len(plans[0]["ops"])
            arr = (L.Op * (g + nops * g + g))()
            n_slots = [[] for _ in range(g)]           # (array index, slot in i[]) that take this chunk's n
            for j in range(g):
                arr[j].stream = j
                for i, rec in enumerate(plans[j]["ops"]):
                    k = g + i * g + j
                    C.memmove(C.addressof(arr[k]), C.addressof(rec["op"]), C.sizeof(L.Op))
                    arr[k].stream = j
                    if rec["n_slot"] is not None:
                        n_slots[j].append((k, rec["n_slot"]))
                last: Act = plans[j]["last"]
                k = g + nops * g + j
                arr[k].kind, arr[k].stream = L.OP_AVGPOOL, j
                arr[k].ptr[0] = last.t.data_ptr()
                for q, v in enumerate((0, last.h, last.w, last.c, last.halo)):
                    arr[k].i[q] = v
                n_slots[j].append((k, 0))
            cache[g] = (arr, n_slots, nops)
        return cache[g]

    # ------------------------------------------------------------------ backbone
    def _run_backbone(self, n_total: int, fill_x0, want_bf16: bool, chunk_head=None, T: int = 1):
        """fill_x0(op, c0, n, x0_ptr, in_halo) turns `op` (an L.Op) into the launch that writes frames
        [c0, c0+n) into the packed stem input at x0_ptr.

        chunk_head(feats, feats_b, c0, n, stream), when given, is called once per chunk right after the chunk's
        launches have been issued, with that chunk's stream current: chunks then hold whole clips of T frames, and a
        chunk's temporal encoder runs on its own stream under the other streams' backbone launches instead of after
        the whole batch.

        Chunks alternate between `self.streams` HIP streams (each with its own workspace), so one
        chunk's kernels fill the CUs another chunk's kernel leaves idle in its last, partial round of
        workgroups and during its per-tile load/store phases.  Each group of chunks is issued by ONE
        asl_run_ops call, op by op across the group, so the host pays one foreign call per group and the
        second stream starts a launch after the first."""
        F = self.spec.feat_dim
        pm = self.parity_mode
        Fp = 2 * F if pm else F        # parity mode: means of the hi and the lo halves side by side, folded below
        feats = torch.empty(n_total, Fp, dtype=torch.float32, device=self.device)
        # bf16 copy of the features = GEMM operand of the temporal encoder; the two-stream model appends
        # the keypoint-MLP output to each row, so the row pitch may exceed F
        # and the BiLSTM takes split-precision features (hi | lo halves) for its input projection
        ld_b = F + (self.spec.kp_hidden if self.spec.temporal == "transformer" else 0)
        lo_off = 0
        if self.spec.temporal == "bilstm":
            ld_b, lo_off = 2 * F, F
        feats_b = torch.empty(n_total, ld_b, dtype=torch.bfloat16, device=self.device) if want_bf16 else None
        nstreams = max(1, min(self.streams, (n_total + self.split_frames - 1) // self.split_frames))
        step = min(self.frames_per_chunk, (n_total + nstreams - 1) // nstreams)
        if chunk_head is not None:
            step = (step + T - 1) // T * T              # whole clips per chunk
        plans = self._get_plans(step, nstreams)
        main = torch.cuda.current_stream()
        side = self._side_streams(nstreams)
        if nstreams > 1:
            for s_ in side:
                s_.wait_stream(main)
            sarr = (C.c_void_p * nstreams)(*[s_.cuda_stream for s_ in side])
        else:
            sarr = (C.c_void_p * 1)(main.cuda_stream)
        starts = list(range(0, n_total, step))
        fbase, bbase = feats.data_ptr(), (feats_b.data_ptr() if feats_b is not None else 0)
        for g0 in range(0, len(starts), nstreams):
            g = min(nstreams, len(starts) - g0)
            arr, n_slots, nops = self._group_array(plans, g)
            for j in range(g):
                c0 = starts[g0 + j]
                n = min(step, n_total - c0)
                plan = plans[j]
                direct = fill_x0(arr[j], c0, n, plan["x0"].data_ptr(), plan["in_halo"])
                sp = plan["stem_planar"]
                if sp is not None:
                    # the stem launch of this chunk: on the packed input the fill op writes, or (fill_x0 returned the
                    # address of the chunk's bf16 clips: the caller's tensor, or what its fill op writes) on those directly
                    k = g + sp["index"] * g + j
                    rec = sp if direct else plan["ops"][sp["index"]]
                    C.memmove(C.addressof(arr[k]), C.addressof(rec["op"]), C.sizeof(L.Op))
                    arr[k].stream = j
                    if direct:
                        arr[k].ptr[0] = direct
                        sp["params"].n = n
                elif direct:
                    raise RuntimeError("this plan has no launch that reads clips in place")
                for rec in plan["ops"]:
                    if rec["params"] is not None:
                        rec["params"].n = n
                for k, slot in n_slots[j]:
                    arr[k].i[slot] = n
                avg = arr[g + nops * g + j]
                avg.ptr[1] = fbase + c0 * Fp * 4
                avg.ptr[2] = (bbase + c0 * ld_b * 2) if (feats_b is not None and not pm) else None
                avg.i[5], avg.i[6] = ld_b, lo_off
            L.run_ops(arr, len(arr), sarr, nstreams, "backbone")
            if chunk_head is not None:
                for j in range(g):
                    c0 = starts[g0 + j]
                    s_ = side[j] if nstreams > 1 else main
                    with torch.cuda.stream(s_):
                        chunk_head(feats, feats_b, c0, min(step, n_total - c0), C.c_void_p(s_.cuda_stream))
        if nstreams > 1:
            for s_ in side:
                main.wait_stream(s_)
        if pm:
            feats = feats[:, :F] + feats[:, F:]
            if feats_b is not None:
                hi = feats.to(torch.bfloat16)
                feats_b[:, :F] = hi
                if lo_off:
                    feats_b[:, lo_off:lo_off + F] = (feats - hi.float()).to(torch.bfloat16)
        return feats, feats_b

    def _side_streams(self, n):
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.device))
        return self._streams[:n]

    def _get_plans(self, nmax: int, count: int):
        if self._plan is None or self._plan[0]["nmax"] < nmax or len(self._plan) < count:
            self._plan = [self._build_plan(nmax) for _ in range(count)]
            self._groups = {}
        return self._plan

    def _heads_per_chunk(self) -> bool:
        """BiLSTM / transformer heads run chunk by chunk on the chunks' streams (see _run_backbone); parity mode folds
        its feature halves after the whole batch and keeps the one-head form."""
        return self.chunk_heads and self.spec.temporal not in ("mean", "i3d") and not self.parity_mode

    def _forward_chunk_heads(self, B: int, T: int, keypoints, run):
        """run(chunk_head, T) drives the backbone; each chunk's clips go through the temporal encoder and head on the
        chunk's stream and land in their rows of the batch's logits."""
        logits = torch.empty(B, self.spec.num_classes, dtype=torch.float32, device=self.device)
        if self.spec.keypoints and keypoints is not None:
            keypoints = keypoints.to(self.device)
        keep = []

        def chunk_head(feats, feats_b, c0, n, stream):
            b0, nb = c0 // T, n // T
            kp = None if keypoints is None else keypoints[b0:b0 + nb]
            out = self._temporal(feats[c0:c0 + n], feats_b[c0:c0 + n], feats_b.shape[1], nb, T, kp, stream)
            logits[b0:b0 + nb].copy_(out)               # on the chunk's stream (current here)
            keep.append((out, self._temporal._keep))
        feats, feats_b = run(chunk_head, T)
        self._keep_heads = (keep, feats, feats_b)       # asynchronous launches: buffers live until the next call
        return logits

    def frame_features(self, clips: torch.Tensor, want_bf16: bool = False, chunk_head=None, T: int = 1):
        """clips [B,T,3,S,S] -> per-frame features fp32 [B*T, F] (and their bf16 copy)."""
        clips = self._check_clips(clips)
        B, T, _, S, _ = clips.shape
        planar = self._planar_input()
        if planar and clips.dtype != torch.bfloat16:
            clips = clips.to(torch.bfloat16)            # the rounding the packed copy would apply (nearest even)
        esz = clips.element_size()
        in_f32 = int(clips.dtype == torch.float32)
        base = clips.data_ptr()

        def fill(op, c0, n, x0_ptr, in_halo):
            if planar:                                    # the stem kernel reads the clips where they lie
                op.kind = L.OP_NOP
                return base + c0 * 3 * S * S * 2
            op.kind = L.OP_PACK_CLIPS
            op.ptr[0], op.ptr[1] = base + c0 * 3 * S * S * esz, x0_ptr
            op.i[0], op.i[1], op.i[2], op.i[3] = in_f32, n, S, in_halo
        return self._run_backbone(B * T, fill, want_bf16, chunk_head, T)

    def _planar_input(self) -> bool:
        """The stem reads bf16 clips [n][3][S][S] in place (the conditions under which _build_plan lays down that launch).
        Constant for a model instance: its plans' input buffers then only ever hold the planar form (uint8 frames are
        resized straight into it), otherwise only the packed one."""
        return self.planar_stem and self.spec.backbone in RESNET_LAYERS and self.fused_stem and not self.parity_mode

    def _check_clips(self, clips):
        if clips.dim() != 5 or clips.shape[2] != 3 or clips.shape[3] != clips.shape[4]:
            raise ValueError(f"clips must be [B,T,3,S,S], got {tuple(clips.shape)}")
        if clips.shape[3] != self.spec.size:
            raise ValueError(f"clips are {clips.shape[3]}x{clips.shape[3]}, spec wants {self.spec.size}")
        if clips.dtype not in (torch.float32, torch.bfloat16):
            clips = clips.float()
        return clips.to(self.device).contiguous()

    # ------------------------------------------------------------------ forward
    def _head(self, feats: torch.Tensor, feats_b, B: int, T: int, keypoints):
        spec = self.spec
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if spec.temporal == "mean":
            F = spec.feat_dim
            pooled = torch.empty(B, F, dtype=torch.float32, device=self.device)
            L.check(L.lib.asl_mean_t_f32(_ptr(feats), B, T, F, _ptr(pooled), stream), "mean_t")
            logits = torch.empty(B, spec.num_classes, dtype=torch.float32, device=self.device)
            L.check(L.lib.asl_head_f32(_ptr(pooled), B, F, _ptr(self._fc_w), _ptr(self._fc_b), spec.num_classes,
                                       _ptr(logits), stream), "head")
            return logits
        return self._temporal(feats, feats_b, feats_b.shape[1], B, T, keypoints, stream)

    # ------------------------------------------------------------------ I3D
    def _run_i3d(self, B: int, T: int, fill_frames, fill_chunk=None):
        """fill_frames(f0, n, dst_ptr, stream) writes frames [f0, f0+n) (flattened over B*T) as packed
        [n][S+6][S+6][4] at dst_ptr; fill_chunk(b0, n, x0_ptr, stream), when given, packs clips [b0, b0+n) into a plan's
        whole input tensor with one launch instead.  Clips run through the 3-D backbone a chunk (16 clips of 64 frames) at a time, the
        Inception branches of a chunk on three streams (I3dPlan.sched).  A batch of two or more full chunks runs them in
        pairs side by side, each on a main stream of its own plus two branch streams, launches issued alternately: one
        chunk's memory-bound pools then run under the other's convolutions (32 x 64: 12.6 -> 12.0 ms; with chunks smaller
        than 16 clips the pairing loses more to under-filled launches than it gains: 16 x 64 as 8 + 8 takes 6.55 ms
        against 6.32)."""
        from .i3d import I3dPlan
        spec, dev, S = self.spec, self.device, self.spec.size
        per = max(1, self.frames_per_chunk // T)
        G = 2 if self.streams > 1 and B >= 2 * per else 1
        nmax = min(per, (B + G - 1) // G)
        plans = self._i3d_plans
        if len(plans) < G or plans[0].T != T or plans[0].nmax < nmax:
            plans = self._i3d_plans = [I3dPlan(spec, self._w, nmax, T, dev) for _ in range(G)]
            self._plan = plans[0]
        main = torch.cuda.current_stream()
        last = plans[0].last
        D = last.d
        frame_feats = torch.empty(B * D, last.c, dtype=torch.float32, device=dev)
        plane = (S + 6) * (S + 6) * 4
        if self.streams <= 1:
            groups = [(main, [main, main, main])]
        else:
            side = self._side_streams(5)
            groups = [(main, [main, side[0], side[1]]), (side[2], [side[2], side[3], side[4]])]
            if not self._i3d_events:
                self._i3d_events = [torch.cuda.Event() for _ in range(8)]
        ev = self._i3d_events
        chunks = [(b0, min(nmax, B - b0)) for b0 in range(0, B, nmax)]
        for c0 in range(0, len(chunks), G):
            act = [(plans[g], groups[g], chunks[c0 + g]) for g in range(min(G, len(chunks) - c0))]
            if G > 1:                                      # the second group starts once the inputs are there
                ev[6].record(main)
                groups[1][0].wait_event(ev[6])
            for plan, (gm, _), (b0, n) in act:
                h = C.c_void_p(gm.cuda_stream)
                if fill_chunk is not None:
                    fill_chunk(b0, n, C.c_void_p(plan.x0.data_ptr()), h)
                    continue
                for b in range(n):  # frames of clip b sit behind 3 zero halo frames of its own padded depth axis
                    dst = plan.x0.data_ptr() + ((b * (T + 6) + 3) * plane) * 2
                    fill_frames((b0 + b) * T, T, C.c_void_p(dst), h)
            hs = [[C.c_void_p(s_.cuda_stream) for s_ in st] for _, (_, st), _ in act]
            for e in plans[0].sched:                       # every plan has the same schedule
                for gi, (plan, (gm, st), (b0, n)) in enumerate(act):
                    if e[0] == "op":
                        plan.ops[e[1]](n, hs[gi][e[2]])
                    elif self.streams <= 1:
                        continue
                    elif e[0] == "fork":
                        ev[3 * gi].record(gm)
                        st[1].wait_event(ev[3 * gi])
                        st[2].wait_event(ev[3 * gi])
                    else:
                        ev[3 * gi + 1].record(st[1])
                        ev[3 * gi + 2].record(st[2])
                        gm.wait_event(ev[3 * gi + 1])
                        gm.wait_event(ev[3 * gi + 2])
            for plan, (gm, _), (b0, n) in act:
                lt = plan.last
                L.check(L.lib.asl_avgpool_nhwc(_ptr(lt.t), n * D, lt.h, lt.w, lt.c, lt.hh,
                                               C.c_void_p(frame_feats.data_ptr() + b0 * D * lt.c * 4), None, 0, 0,
                                               C.c_void_p(gm.cuda_stream)), "avgpool")
            if len(act) > 1:
                ev[7].record(groups[1][0])
                main.wait_event(ev[7])
        stream = C.c_void_p(main.cuda_stream)
        pooled = torch.empty(B, last.c, dtype=torch.float32, device=dev)
        L.check(L.lib.asl_window_mean_t_f32(_ptr(frame_feats), B, D, last.c, min(2, D), _ptr(pooled), stream), "window_mean")
        logits = torch.empty(B, spec.num_classes, dtype=torch.float32, device=dev)
        L.check(L.lib.asl_head_f32(_ptr(pooled), B, last.c, _ptr(self._fc_w), _ptr(self._fc_b), spec.num_classes,
                                   _ptr(logits), stream), "head")
        return logits

    @torch.no_grad()
    def forward(self, clips: torch.Tensor, keypoints: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self._w:
            raise RuntimeError("load_state_dict() first (or construct with seed=...)")
        B, T = clips.shape[:2]
        if self.spec.backbone == "i3d":
            clips = self._check_clips(clips)
            S, esz, in_f32, base = self.spec.size, clips.element_size(), int(clips.dtype == torch.float32), clips.data_ptr()

            def fill(f0, n, dst, stream):
                L.check(L.lib.asl_pack_clips(C.c_void_p(base + f0 * 3 * S * S * esz), in_f32, n, S, 3, dst, stream),
                        "pack_clips")
            def fill_chunk(b0, n, dst, stream):
                L.check(L.lib.asl_pack_clips_3d(C.c_void_p(base + b0 * T * 3 * S * S * esz), in_f32, n, T, S, 3, 3, dst, stream),
                        "pack_clips_3d")
            return self._run_i3d(B, T, fill, fill_chunk)
        if self.spec.keypoints and keypoints is None:
            raise ValueError(f"spec {self.spec.name} needs keypoints [B,T,21,2]")
        if self._heads_per_chunk():
            return self._forward_chunk_heads(B, T, keypoints, lambda ch, t: self.frame_features(clips, True, ch, t))
        feats, feats_b = self.frame_features(clips, want_bf16=self.spec.temporal != "mean")
        return self._head(feats, feats_b, B, T, keypoints)

    __call__ = forward

    @torch.no_grad()
    def classify_clip(self, frames: torch.Tensor, keypoints: Optional[torch.Tensor] = None, topk: int = 5):
        """frames uint8 [T,H0,W0,3] or [B,T,H0,W0,3] -> (idx int64 [B,k], prob float32 [B,k])."""
        if frames.dtype != torch.uint8:
            raise ValueError("frames must be uint8")
        if frames.dim() == 4:
            frames = frames[None]
            if keypoints is not None and keypoints.dim() == 3:
                keypoints = keypoints[None]
        if frames.dim() != 5 or frames.shape[-1] != 3:
            raise ValueError(f"frames must be [B,T,H0,W0,3], got {tuple(frames.shape)}")
        frames = frames.to(self.device).contiguous()
        B, T, H0, W0, _ = frames.shape
        S = self.spec.size
        mean = (C.c_float * 3)(*IMAGENET_MEAN)
        std = (C.c_float * 3)(*IMAGENET_STD)
        base = frames.data_ptr()
        planar = self._planar_input()

        def fill(op, c0, n, x0_ptr, in_halo):
            op.kind = L.OP_PREPROCESS
            op.ptr[0], op.ptr[1], op.ptr[2] = base + c0 * H0 * W0 * 3, x0_ptr, None
            op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5] = n, H0, W0, S, in_halo, int(planar)
            for q in range(3):
                op.f[q], op.f[3 + q] = IMAGENET_MEAN[q], IMAGENET_STD[q]
            # planar: the chunk's bf16 clips [n][3][S][S] are written at the head of the plan's input buffer (larger
            # than they are) and the stem reads them from there
            return x0_ptr if planar else None
        if self.spec.backbone == "i3d":
            def fill3(f0, n, dst, stream):
                L.check(L.lib.asl_preprocess_u8(C.c_void_p(base + f0 * H0 * W0 * 3), n, H0, W0, S, 3, mean, std, dst,
                                                None, stream), "preprocess")
            logits = self._run_i3d(B, T, fill3)
        elif self._heads_per_chunk():
            logits = self._forward_chunk_heads(B, T, keypoints, lambda ch, t: self._run_backbone(B * T, fill, True, ch, t))
        else:
            feats, feats_b = self._run_backbone(B * T, fill, self.spec.temporal != "mean")
            logits = self._head(feats, feats_b, B, T, keypoints)
        k = int(topk)
        idx = torch.empty(B, k, dtype=torch.int64, device=self.device)
        prob = torch.empty(B, k, dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.asl_softmax_topk(_ptr(logits), B, self.spec.num_classes, k, _ptr(idx), _ptr(prob), stream),
                "softmax_topk")
        return idx, prob

    @torch.no_grad()
    def preprocess(self, frames: torch.Tensor) -> torch.Tensor:
        """uint8 [B,T,H0,W0,3] -> normalised fp32 clips [B,T,3,S,S] (what forward() takes)."""
        frames = frames.to(self.device).contiguous()
        B, T, H0, W0, _ = frames.shape
        S = self.spec.size
        out = torch.empty(B, T, 3, S, S, dtype=torch.float32, device=self.device)
        scratch = torch.empty(B * T, S, S, 4, dtype=torch.bfloat16, device=self.device)
        mean = (C.c_float * 3)(*IMAGENET_MEAN)
        std = (C.c_float * 3)(*IMAGENET_STD)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.asl_preprocess_u8(_ptr(frames), B * T, H0, W0, S, 0, mean, std, _ptr(scratch), _ptr(out),
                                        stream), "preprocess")
        return out

    def capture(self, B: int, T: int) -> "CapturedForward":
        """forward() for one batch shape as a HIP graph: see CapturedForward."""
        return CapturedForward(self, B, T)

    def workspace_bytes(self) -> int:
        if self._plan is None:
            return 0
        if isinstance(self._plan, list):
            return sum(p["pool"].nbytes() + p["x0"].numel() * 2 for p in self._plan)
        return sum(p.nbytes() for p in self._i3d_plans)


class CapturedForward:
    """model.forward for one (B, T), captured once into a HIP graph (chunk streams, branch streams, their events and the
    recurrence's per-step launches included) and replayed per call: the host issues one graph launch instead of every
    kernel.  Dependent kernels start no sooner inside a graph than behind one another on a stream, so this pays where
    issuing the launches costs the host about as long as the GPU runs them (I3D 1 x 64: 1.07 -> 0.99 ms), not at
    throughput batch sizes.  Inputs are copied into the captured buffers; the returned logits are the captured output
    tensor, overwritten by the next call."""

    def __init__(self, model: "AslClipModel", B: int, T: int):
        spec, dev = model.spec, model.device
        self.model = model
        self.clips = torch.zeros(B, T, 3, spec.size, spec.size, dtype=torch.bfloat16, device=dev)
        self.keypoints = torch.zeros(B, T, 21, 2, dtype=torch.float32, device=dev) if spec.keypoints else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # plans, streams and LDS-size attributes come into being outside the capture
            for _ in range(2):
                model(self.clips, self.keypoints)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logits = model(self.clips, self.keypoints)

    @torch.no_grad()
    def __call__(self, clips: torch.Tensor, keypoints: Optional[torch.Tensor] = None) -> torch.Tensor:
        if tuple(clips.shape) != tuple(self.clips.shape):
            raise ValueError(f"captured for clips {tuple(self.clips.shape)}, got {tuple(clips.shape)}")
        if (self.keypoints is None) != (keypoints is None):
            raise ValueError("keypoints: the captured model " + ("takes" if self.keypoints is not None else "does not take") + " them")
        self.clips.copy_(clips)
        if keypoints is not None:
            self.keypoints.copy_(keypoints)
        self.graph.replay()
        return self.logits


def init_state_dict(spec: ModelSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init parameters with the names/shapes of `spec` (bench and smoke use this when no
    checkpoint is given): He-normal convs, non-trivial BatchNorm statistics, uniform linears; all
    values bf16-representable."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in param_shapes(spec).items():
        if name.endswith("running_var") or (".bn" in name or "downsample.1" in name or ".norm" in name) and name.endswith(".weight"):
            t = torch.empty(shp).uniform_(0.5, 1.5, generator=g)
        elif name.endswith("running_mean") or name.endswith(".bias") and len(shp) == 1 and (
                ".bn" in name or "downsample.1" in name or ".norm" in name):
            t = torch.empty(shp).normal_(0.0, 0.1, generator=g)
        elif len(shp) >= 4:
            fan_out = shp[0]
            for d in shp[2:]:
                fan_out *= d
            t = torch.empty(shp).normal_(0.0, (2.0 / fan_out) ** 0.5, generator=g)
        elif len(shp) == 2:
            k = 1.0 / (shp[1] ** 0.5)
            t = torch.empty(shp).uniform_(-k, k, generator=g)
        else:
            t = torch.empty(shp).uniform_(-0.05, 0.05, generator=g)
        sd[name] = t.to(torch.bfloat16).to(torch.float32)
    return sd
