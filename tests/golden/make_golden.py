"""Generates the golden fixtures under tests/golden/ from the oracle (run from the repo root:
``python tests/golden/make_golden.py``).  Each .npz holds the seeds and shapes needed to
regenerate its input, the input's sha256, and the oracle's outputs.  The reference ships no
vectors of its own, so these pin the build-owned oracle, not the reference."""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import asl_oracle as O  # noqa: E402


def sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def forward_case(name, B, T, S, seed, heavy=0):
    m = O.build(name, seed=seed)
    x = O.make_clips(B, T, S, seed)
    kp = O.make_keypoints(B, T, seed) if name == "twostream_transformer" else None
    with torch.no_grad():
        y = m(x, kp)
    np.savez(os.path.join(HERE, f"{name}_fwd_b{B}t{T}s{S}.npz"), model=name, kind="forward", B=B, T=T, S=S,
             seed=seed, heavy=heavy, input_sha256=sha(x), logits=y.numpy())
    print(name, "forward", B, T, S, "logit std %.3f" % y.std())


def classify_case(name, B, T, S, H0, W0, seed, topk=5):
    m = O.build(name, seed=seed)
    fr = O.make_frames(B, T, H0, W0, seed)
    kp = O.make_keypoints(B, T, seed) if name == "twostream_transformer" else None
    idx, prob = O.classify_clip(m, fr, kp, topk=topk, size=S)
    np.savez(os.path.join(HERE, f"{name}_cls_b{B}t{T}s{S}.npz"), model=name, kind="classify", B=B, T=T, S=S,
             H0=H0, W0=W0, seed=seed, heavy=0, topk=topk, input_sha256=sha(fr), idx=idx.numpy(), prob=prob.numpy())
    print(name, "classify", idx[0].tolist())


def full_length_cases():
    """the configurations' own sequence lengths (minutes of CPU time): `python make_golden.py full`"""
    forward_case("resnet50_bilstm", 1, 64, 224, 0, heavy=1)
    forward_case("twostream_transformer", 1, 128, 224, 0, heavy=1)
    forward_case("i3d", 2, 64, 224, 0, heavy=1)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if sys.argv[1:] == ["full"]:
        full_length_cases()
        sys.exit(0)
    forward_case("tiny", 2, 8, 112, 0)
    classify_case("tiny", 2, 8, 112, 60, 80, 0)
    forward_case("resnet18_mean", 2, 8, 224, 0)
    forward_case("resnet18_mean", 1, 2, 64, 1)
    classify_case("resnet18_mean", 1, 4, 224, 120, 160, 0)
    forward_case("resnet50_bilstm", 2, 8, 224, 0, heavy=1)
    forward_case("twostream_transformer", 2, 16, 224, 0, heavy=1)
    forward_case("i3d", 1, 16, 224, 0, heavy=1)
    forward_case("i3d", 1, 32, 224, 0, heavy=1)
    forward_case("i3d", 1, 64, 224, 0, heavy=1)
