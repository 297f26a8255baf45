#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum and per-launch mean of every
counter.  HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies
128-byte read requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact.

python tools/pmc_summary.py [--json out.json --frames-per-chunk N] a.csv b.csv ...
With --json, also writes the figures bench.py reports next to its timing: HBM bytes of the headline path's kernels
per chunk of frames -- the stem launch that reads the clips in place (once per chunk) plus, for the kernels the
uint8-frames leg of the bench shares (blocks, convolutions, pools, head), their bytes per launch x launches per chunk
over both legs; the kernels only that leg runs (resize, softmax; a packed-input stem and pack when ASL_PLANAR_STEM=0) are
left out --
and the MFMA-busy ratio of the conv kernels, SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x busy cycles), busy
cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import csv
import json
import sys
from collections import defaultdict

CONV = ("stem_pool_kernel", "stem_strip_kernel", "block_c64_pair_kernel", "block_c64_kernel", "conv3x3_c64_kernel", "conv3x3_pp_kernel", "conv3x3_wide_kernel", "conv3x3_patch_kernel",
        "gather_gemm_kernel")


def main(argv):
    json_out, fpc = None, None
    paths = []
    it = iter(argv)
    for a in it:
        if a == "--json":
            json_out = next(it)
        elif a == "--frames-per-chunk":
            fpc = int(next(it))
        else:
            paths.append(a)
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0][-96:]
                c = row["Counter_Name"]
                agg[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    print("kernel,counter,launches,sum,per_launch")
    hbm = {}
    for k in sorted(agg):
        for c in sorted(agg[k]):
            n = calls[k][c]
            print(f"{k},{c},{n},{agg[k][c]:.6g},{agg[k][c] / n:.6g}")
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            # the two counters come from separate passes of the same command: scale each to a per-launch figure
            per = 2 * agg[k]["FETCH_SIZE"] / calls[k]["FETCH_SIZE"] + agg[k]["WRITE_SIZE"] / calls[k]["WRITE_SIZE"]
            n = calls[k]["FETCH_SIZE"]
            hbm[k] = (per * 1024, n)
            print(f"{k},HBM_BYTES_corrected,{n},{per * 1024 * n:.6g},{per * 1024:.6g}")
    if json_out:
        stem = [k for k in hbm if "stem_strip_kernel<true>" in k]          # the headline path's stem
        stems_all = [k for k in hbm if "stem_strip_kernel" in k or "stem_pool_kernel" in k]
        other_leg = ("stem_strip_kernel<false>", "stem_pool_kernel", "preprocess_kernel", "softmax_topk_kernel", "pack_clips")
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from video_asl_recognition_amd._lib import source_hash
        # stamp: the kernel sources these counters were collected on (bench.py drops the figures when it differs)
        out = {"frames_per_chunk": fpc, "source_hash": source_hash(), "per_kernel": {}}
        if stem:
            chunks = hbm[stem[0]][1]
            chunks_all = sum(hbm[k][1] for k in stems_all)
            tot = hbm[stem[0]][0]
            for k, (b, n) in hbm.items():
                if "asl_detail" in k and k != stem[0] and not any(o in k for o in other_leg):
                    tot += b * n / chunks_all
            out["hbm_bytes_per_chunk"] = tot
            out["chunks_profiled"] = chunks
        busy = mf = 0.0
        for k in agg:
            rec = {}
            if k in hbm:
                rec["hbm_bytes_per_launch"] = hbm[k][0]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and "GRBM_GUI_ACTIVE" in agg[k]:
                cyc = agg[k]["GRBM_GUI_ACTIVE"] / 8.0
                rec["mfma_busy"] = agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else None
                if any(c in k for c in CONV):
                    busy += cyc
                    mf += agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
            if rec and "asl_detail" in k:
                out["per_kernel"][k] = rec
        if busy:
            out["mfma_busy"] = round(mf / (1024.0 * busy), 4)
            out["mfma_busy_note"] = ("conv kernels, each profiled launch alone on the GPU (counter collection serialises "
                                     "kernels): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
        with open(json_out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
