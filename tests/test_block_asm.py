"""CPU: the hand-written MFMA loops of the fused BasicBlock kernels, checked in the compiler's output.

The K loops in csrc/block_mfma.h issue `ds_read_b128` and `v_mfma_f32_32x32x16_bf16` from inline assembly with hand-counted
`s_waitcnt lgkmcnt(N)`.  The compiler treats a read's destination as ready at once and cannot see the MFMA inside the
asm text, so three things would silently break the arithmetic and only show in numeric GPU tests:
  * a copy / spill of a fragment or accumulator register placed between a read and its MFMA, or behind an MFMA
    without the wait states its result needs (v_mov, v_accvgpr_*, scratch_*);
  * a read whose destination is the B fragment of the MFMA issued just before it (an MFMA that finds the pipe busy
    reads its operands when it starts, up to 32 cycles after it issues);
  * a tile whose last MFMA is not followed by the 18+ wait states before anything reads the accumulators.
This test compiles the default (paired) kernel to assembly (no GPU needed) and checks every 36-step tile for them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video_asl_recognition_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def _regs(text):
    """set of ('v'|'a', index) named in an operand string"""
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _instructions(path):
    ins = []
    with open(path) as f:
        for line in f:
            s = line.split(";")[0].strip()
            if s.endswith(":") and not s.startswith("."):
                continue                            # symbol of the function itself
            if s.endswith(":"):
                ins.append(("LABEL", s))            # basic-block boundary
                continue
            if not s or s.startswith("."):
                continue
            parts = s.split(None, 1)
            ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return ins


def _tiles(ins):
    """(index of first MFMA, index of last MFMA) per tile of 36 MFMAs; a tile's last MFMA is the one followed by `s_nop 15`"""
    idx = [i for i, (op, _) in enumerate(ins) if op == "v_mfma_f32_32x32x16_bf16"]
    assert idx and len(idx) % 36 == 0, f"{len(idx)} MFMAs: not whole 36-step tiles"
    tiles, first, count = [], None, 0
    for i in idx:
        first = i if first is None else first
        count += 1
        if ins[i + 1] == ("s_nop", "15"):
            assert count == 36, f"{count} MFMAs before a tile's end"
            tiles.append((first, i))
            first, count = None, 0
    assert first is None, "MFMAs behind the last fenced tile end"
    return tiles


# block_c64.hip (the one-wave fallback kernel) shares block_mfma.h but keeps its tiles in a rotated loop with the previous
# tile's epilogue sliced in between: its text is not straight-line per tile, so only the default kernel is checked here.
@pytest.mark.parametrize("src", ["block_c64_pair.hip"])
def test_block_kernel_k_loops(tmp_path, src):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not present")
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    ins = _instructions(out)
    tiles = _tiles(ins)
    assert len(tiles) >= 2
    MFMA = "v_mfma_f32_32x32x16_bf16"
    fenced = 0
    for first, last in tiles:
        acc, prev_b = set(), None
        for i in range(first, last + 1):
            op, args = ins[i]
            if op == MFMA:
                ops = [a.strip() for a in args.split(",")]
                b = _regs(ops[2])
                # walk back to the instruction that filled this B fragment: it must be one ds_read_b128 of exactly these
                # registers, and nothing between it and this MFMA may touch them (no copy, no spill, no reuse)
                j = i - 1
                while j >= 0 and i - j < 1500:
                    pop, pargs = ins[j]
                    if pop == "LABEL" or pop.startswith("s_cbranch") or pop == "s_branch":
                        break           # the fragment crosses a basic-block boundary (a loop's back edge carries the
                                        # reads of a tile's first steps): textual order says nothing beyond here
                    if _regs(pargs) & b:
                        assert pop == "ds_read_b128" and _regs(pargs.split(",")[0]) == b, \
                            f"{src}: between a fragment's read and its MFMA ({args}): {pop} {pargs}"
                        break
                    j -= 1
                acc |= _regs(ops[0])
                prev_b = b
                continue
            assert not op.startswith("scratch_"), f"{src}: spill inside a K loop: {op} {args}"
            if op == "ds_read_b128":
                dst = _regs(args.split(",")[0])
                assert not (dst & acc), f"{src}: read into an accumulator: {args}"
                assert prev_b is None or not (dst & prev_b), \
                    f"{src}: read refills the fragment of the MFMA just issued: ds_read_b128 {args}"
                continue
            if op in ("s_nop", "s_waitcnt", "s_setprio", "s_barrier", "LABEL"):
                continue
            # other work is scheduled between the MFMAs (the next tile's addresses; in the one-wave kernel, slices of
            # the previous tile's epilogue): allowed as long as it leaves the accumulators alone
            assert not (_regs(args) & acc), f"{src}: {op} {args} touches an accumulator inside the K loop"
        # the statement of the last MFMA carries its own wait states: 16 + 4 of them directly behind it
        tail = [ins[last + 1], ins[last + 2]]
        if tail[0] == ("s_nop", "15"):
            assert tail[1] == ("s_nop", "3"), f"{src}: last MFMA of a tile not fenced: {tail}"
            fenced += 1
    assert fenced * 36 == sum(1 for op, _ in ins if op == MFMA), f"{src}: {fenced} fenced tile ends"
