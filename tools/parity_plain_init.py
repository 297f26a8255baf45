#!/usr/bin/env python3
"""Reports (does not gate) the path's error against the CPU oracle under the "plain" init recipe
(gamma ~ U(0.5, 1.5) in every BatchNorm, fan-out He init everywhere), next to the conditioned recipe the
parity tests use, in the default mode and (ResNet backbones) in parity mode, where every activation travels as a
bf16 hi + lo pair.  One line per (model, recipe, mode): rel-L2 of the logits, arg-max agreement, logit std.
python tools/parity_plain_init.py [json-out]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import asl_oracle as O  # noqa: E402
from video_asl_recognition_amd import AslClipModel  # noqa: E402

CASES = [("resnet18_mean", 1, 8), ("resnet18_mean", 1, 64), ("resnet50_bilstm", 1, 8), ("i3d", 1, 16)]


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = []
    for name, B, T in CASES:
        for recipe in ("conditioned", "plain"):
            om = O.build(name, seed=0, recipe=recipe)
            ref = None
            for parity_mode in ((False, True) if name != "i3d" else (False,)):
                m = AslClipModel(name, frames_per_chunk=64, parity_mode=parity_mode)
                m.load_state_dict(om.state_dict())
                clips = O.make_clips(B, T, m.spec.size, 0)
                if ref is None:
                    with torch.no_grad():
                        ref = om(clips)
                y = m(clips.cuda()).cpu()
                rel = float((y - ref).norm() / ref.norm())
                rec = dict(model=name, B=B, T=T, recipe=recipe, mode="parity" if parity_mode else "default", rel_l2=rel,
                           argmax_identical=bool(torch.equal(y.argmax(1), ref.argmax(1))),
                           logit_std=float(ref.std()), finite=bool(torch.isfinite(y).all()))
                out.append(rec)
                print(json.dumps(rec), flush=True)
                del m
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
