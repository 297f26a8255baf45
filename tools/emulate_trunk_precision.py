#!/usr/bin/env python3
"""CPU experiment (no GPU): how much of the path's error under the "plain" init recipe is due to the bf16
residual trunk?  Runs the oracle's ResNet-50 + BiLSTM with activation rounding emulated in torch:
  all_bf16    every fused layer output rounded to bf16 (what the kernels do)
  fp32_trunk  block outputs kept in fp32 on the residual path; only conv INPUTS are rounded to bf16
and prints rel-L2 of the logits against the unrounded fp32 oracle.  python tools/emulate_trunk_precision.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import asl_oracle as O  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).float()


def bottleneck(b, x_in, x_res):
    """x_in: bf16-rounded conv input; x_res: residual-path value (rounded or fp32)"""
    idt = x_res if b.downsample is None else b.downsample(x_in)
    if b.downsample is not None:
        idt = rb(idt)                      # the downsample output is a stored bf16 tensor either way
    y = rb(F.relu(b.bn1(b.conv1(x_in))))
    y = rb(F.relu(b.bn2(b.conv2(y))))
    return F.relu(b.bn3(b.conv3(y)) + idt)


@torch.no_grad()
def run(om, clips, mode):
    bb = om.backbone
    x = clips.flatten(0, 1)
    x = rb(bb.maxpool(F.relu(bb.bn1(bb.conv1(x)))))
    res = x
    for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for blk in layer:
            y = bottleneck(blk, x, res)
            x = rb(y)
            res = y if mode == "fp32_trunk" else x
    f = (res if mode == "fp32_trunk" else x).mean((2, 3)).view(clips.shape[0], clips.shape[1], -1)
    return om.temporal_head(f)


def main():
    torch.set_num_threads(8)
    for recipe in ("conditioned", "plain"):
        om = O.build("resnet50_bilstm", seed=0, recipe=recipe)
        clips = O.make_clips(1, 8, 224, 0)
        with torch.no_grad():
            ref = om(clips)
        for mode in ("all_bf16", "fp32_trunk"):
            y = run(om, clips, mode)
            print(f"{recipe:12s} {mode:10s} rel-L2 {float((y - ref).norm() / ref.norm()):.3e}  argmax same "
                  f"{bool(torch.equal(y.argmax(1), ref.argmax(1)))}", flush=True)


if __name__ == "__main__":
    main()
