#!/usr/bin/env python3
"""Times every launch of a model's backbone plan on its own (one op at a time, `reps` back-to-back launches between two
events on one stream) for one chunk of frames, and prints per op: kind, shape, time, TFLOP/s of its algorithmic MACs and
GB/s of its algorithmic bytes (input + output + residual, each once, bf16).  Tells which launches of a configuration sit
far from both rooflines.
Usage: python tools/time_plan_ops.py <spec name> [frames = 256] [reps = 20]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_asl_recognition_amd import AslClipModel, _lib as L  # noqa: E402
from video_asl_recognition_amd.spec import SPECS  # noqa: E402

KIND = {v: k for k, v in vars(L).items() if k.startswith("OP_") and isinstance(v, int)}


def describe(rec, n):
    """(text, MACs, bytes) of one plan record at n frames"""
    p, kind = rec["params"], int(rec["op"].kind)
    if kind == L.OP_CONV2D:
        oh = (p.h + 2 * p.pad - p.kh) // p.stride + 1
        ow = (p.w + 2 * p.pad - p.kw) // p.stride + 1
        macs = n * oh * ow * p.cout * p.cin * p.kh * p.kw
        byts = 2 * n * (p.h * p.w * p.cin + oh * ow * p.cout * (2 if p.res_ctot else 1))
        return f"{p.kh}x{p.kw}/{p.stride} {p.cin:4d}->{p.cout:4d} @{p.h}x{p.w}{' +res' if p.res_ctot else ''}", macs, byts
    if kind == L.OP_CONV1X1_DUAL:
        macs = n * p.h * p.w * p.cout * (p.cin + p.cin2)
        byts = 2 * n * p.h * p.w * (p.cin + p.cin2 + p.cout)
        return f"1x1 {p.cin:4d}->{p.cout:4d} @{p.h}x{p.w} + shortcut 1x1/{p.stride2} {p.cin2:4d}->{p.cout:4d}", macs, byts
    if kind == L.OP_CONV1X1_CHAIN:
        macs = n * p.h * p.w * p.cmid * (p.cin + p.cout)
        byts = 2 * n * p.h * p.w * (p.cin + 2 * p.cmid + p.cout)
        return f"1x1 {p.cin:4d}->{p.cmid:4d} +res, then 1x1 ->{p.cout:4d} @{p.h}x{p.w}", macs, byts
    if kind == L.OP_BASIC_BLOCK:
        macs = 2 * n * p.h * p.w * 64 * 64 * 9
        return f"block 64 @{p.h}x{p.w}", macs, 2 * n * p.h * p.w * 64 * 2
    if kind == L.OP_STEM_POOL:
        macs = n * (p.h // 2) * (p.w // 2) * 64 * 147
        return f"stem+pool @{p.h}x{p.w}", macs, 2 * n * (p.h * p.w * 3 + (p.h // 4) * (p.w // 4) * 64)
    if kind == L.OP_BLOCK_ENTRY:
        macs = n * (p.h // 2) * (p.w // 2) * p.cout * p.cin * 10
        return f"entry {p.cin}->{p.cout} @{p.h}x{p.w}", macs, 2 * n * (p.h * p.w * p.cin + (p.h // 2) * (p.w // 2) * p.cout * 2)
    if kind == L.OP_CONV3D:
        od, oh, ow = [(getattr(p, a) + 2 * getattr(p, "p" + a) - getattr(p, "k" + a)) // getattr(p, "s" + a) + 1 for a in "dhw"]
        macs = n * od * oh * ow * p.cout * p.cin * p.kd * p.kh * p.kw
        return (f"{p.kd}x{p.kh}x{p.kw}/{p.sd}{p.sh}{p.sw} {p.cin:4d}->{p.cout:4d} @{p.d}x{p.h}x{p.w}", macs,
                2 * n * (p.d * p.h * p.w * p.cin + od * oh * ow * p.cout))
    return KIND.get(kind, str(kind)), 0, 0


def main():
    name = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    spec = SPECS[name]
    m = AslClipModel(spec, seed=0, frames_per_chunk=frames, streams=1)
    T = spec.frames
    B = max(1, frames // T)
    g = torch.Generator(device="cuda").manual_seed(0)
    clips = torch.randn(B, T, 3, spec.size, spec.size, generator=g, device="cuda").to(torch.bfloat16)
    kp = torch.rand(B, T, 21, 2, generator=g, device="cuda") if spec.temporal == "transformer" else None
    m(clips, kp)                                   # builds the plan and leaves real activations in every buffer
    torch.cuda.synchronize()
    plan = m._plan[0]
    stream = torch.cuda.current_stream()
    sarr = (C.c_void_p * 1)(stream.cuda_stream)
    n = B * T if spec.backbone != "i3d" else B
    total = 0.0
    print(f"# {name}: {B} clips x {T} frames per chunk, one launch at a time, {reps} reps")
    only = os.environ.get("ONLY")          # e.g. ONLY=OP_STEM_POOL: time just the launches of that kind
    sp = plan.get("stem_planar") if m.planar_stem else None
    for idx, rec in enumerate(plan["ops"]):
        if only and KIND.get(int(rec["op"].kind)) != only:
            continue
        arr = (L.Op * 1)()
        if sp is not None and idx == sp["index"]:      # the stem launch that reads the clips in place
            rec = dict(op=sp["op"], params=sp["params"], n_slot=None)
            C.memmove(C.addressof(arr[0]), C.addressof(rec["op"]), C.sizeof(L.Op))
            arr[0].ptr[0] = clips.data_ptr()
        else:
            C.memmove(C.addressof(arr[0]), C.addressof(rec["op"]), C.sizeof(L.Op))
        arr[0].stream = 0
        if rec["params"] is not None:
            rec["params"].n = n
        if rec["n_slot"] is not None:
            arr[0].i[rec["n_slot"]] = n
        for _ in range(2):
            L.run_ops(arr, 1, sarr, 1, "op")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            L.run_ops(arr, 1, sarr, 1, "op")
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        text, macs, byts = describe(rec, n)
        total += us
        print(f"{KIND.get(int(rec['op'].kind), '?'):16s} {text:44s} {us:9.1f} us  {2 * macs / us / 1e6:7.1f} TF/s  {byts / us / 1e3:7.1f} GB/s")
    print(f"# sum {total:.1f} us")


if __name__ == "__main__":
    main()
