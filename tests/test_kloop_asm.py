"""CPU: the K loops of the LDS-resident-patch convolution kernels, checked in the compiler's output.

conv3x3_pp_kernel and stem3d_kernel retire their LDS-DMA weight stages and patch pieces with `s_waitcnt vmcnt(N)`, N = the
pieces issued in the current load phase (csrc/conv3x3_pp.hip, csrc/conv_stem3d.hip).  That holds only while the loop's
vector-memory stream is exactly the DMA instructions the source issues:
  * a register reloaded from scratch inside the loop is one more operation on the same counter (the wait then lets a
    DMA piece through unfinished) -- and the compiler guards the reload's first use with `s_waitcnt vmcnt(0)`;
  * a value reloaded from scratch just AHEAD of the loop makes the compiler put `s_waitcnt vmcnt(0)` in front of its
    first use inside the loop on every iteration, which drains the DMA queue (correct, but the prefetch distance is gone).
Neither gives a compile error or a wrong result in a small test; this test compiles the files to assembly (no GPU needed)
and checks the loops."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video_asl_recognition_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(tmp, src):
    out = tmp / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    return open(out).read()


def _kernel(text, mangled_prefix):
    """instruction list [(mnemonic, operands, label-or-None)] of the first kernel whose symbol starts with the prefix"""
    m = re.search(r"^(" + re.escape(mangled_prefix) + r"\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert m, f"{mangled_prefix} not found in the assembly"
    ins = []
    for line in m.group(2).splitlines():
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            if s.endswith(":"):
                ins.append(("label", s[:-1]))
            continue
        parts = s.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return m.group(1), ins


def _main_loop(ins):
    """instructions of the loop that holds the most MFMAs: from the target label of a backward branch to that branch"""
    labels = {a: i for i, (op, a) in enumerate(ins) if op == "label"}
    best = None
    for i, (op, a) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            t = labels.get(a.strip())
            if t is not None and t < i:
                body = ins[t:i + 1]
                n = sum(1 for o, _ in body if o.startswith("v_mfma"))
                # (block placement may leave part of the body behind the first back edge: take the last one)
                if best is None or (n, len(body)) > (best[0], len(best[1])):
                    best = (n, body)
    assert best, "no loop found"
    return best[1]


def _is_dma(op, a):
    return op.startswith("buffer_load_dword") and a.rstrip().endswith("lds")


def _check_loop(name, loop, n_mfma, waits_allowed):
    assert sum(1 for o, _ in loop if o.startswith("v_mfma")) == n_mfma, f"{name}: the K loop does not hold {n_mfma} MFMAs"
    for op, a in loop:
        assert not op.startswith("scratch_"), f"{name}: scratch access inside the K loop: {op} {a}"
        assert not op.startswith("global_") and not op.startswith("flat_"), f"{name}: {op} {a} inside the K loop"
        if op.startswith("buffer_"):
            assert _is_dma(op, a), f"{name}: a buffer operation that is not an LDS-DMA inside the K loop: {op} {a}"
    waits = [int(re.search(r"vmcnt\((\d+)\)", a).group(1)) for op, a in loop if op == "s_waitcnt" and "vmcnt" in a]
    assert waits, f"{name}: no counted wait in the K loop"
    assert set(waits) <= waits_allowed, f"{name}: wait counts {sorted(set(waits))}, the source issues {sorted(waits_allowed)}"
    return waits


@pytest.fixture(scope="module")
def tmp(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    return tmp_path_factory.mktemp("kloop")


def test_stem3d_k_loop(tmp):
    text = _compile(tmp, "conv_stem3d.hip")
    name, ins = _kernel(text, "_ZN10asl_detail13stem3d_kernel")
    loop = _main_loop(ins)
    # 7 filter rows x 16 MFMAs; the early half waits with one piece in flight, the late half with two (rows 0..3) or none
    waits = _check_loop(name, loop, 7 * 16, {0, 1, 2})
    assert waits.count(1) == 7 and waits.count(2) == 4 and waits.count(0) == 3, f"{name}: waits {waits}"
    # per depth tap: 7 weight pieces (early half) + 8 patch pieces (late half)
    assert sum(1 for o, a in loop if _is_dma(o, a)) == 15, f"{name}: DMA instructions in the loop"
    # the only full drain is the late half's own (it has nothing in flight in rows 4..6): each sits in the same asm
    # statement as the barrier that follows it, i.e. it is never a wait the compiler added in front of an LDS read
    for i, (op, a) in enumerate(loop):
        if op == "s_waitcnt" and re.search(r"vmcnt\(0\)", a):
            assert "lgkmcnt(0)" in a and loop[i + 1][0] == "s_barrier", f"{name}: compiler-inserted s_waitcnt vmcnt(0) in the K loop"


@pytest.mark.parametrize("sym,n_mfma", [("_ZN10asl_detail17conv3x3_pp_kernelILi4ELi1EE", 9 * 16),
                                        ("_ZN10asl_detail17conv3x3_pp_kernelILi4ELi3EE", 9 * 16)], ids=["2d", "3d"])
def test_large_tile_k_loop(tmp, sym, n_mfma):
    text = _compile(tmp, "conv3x3_pp.hip")
    name, ins = _kernel(text, sym)
    loop = _main_loop(ins)
    # nine taps: one weight piece per wave and step, one patch piece more in the first four; prefetch distance 2
    waits = _check_loop(name, loop, n_mfma, {1, 2})
    assert waits.count(2) == 4 and waits.count(1) == 5, f"{name}: waits {waits}"
    assert sum(1 for o, a in loop if _is_dma(o, a)) == 13, f"{name}: DMA instructions in the loop"
