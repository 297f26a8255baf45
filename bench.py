#!/usr/bin/env python3
"""Benchmark of the clip -> logits hot path: ResNet-18 per-frame backbone + mean over T +
Linear(512,100), T=64, 224x224, bf16, synthetic clips resident in HBM.

  python bench.py --gpus N --steps K --warmup W

For N>1 the command may be started either way: under torch.distributed.run (WORLD_SIZE etc. in the
environment: this process is one rank), or bare -- then this process touches no GPU, starts the N
ranks itself through torch.distributed.run on 127.0.0.1, and exits with their return code.

A step = one forward() over one batch of `--clips` clips per GPU (pack -> fused stem + max-pool ->
layer 1 as two fused BasicBlocks -> the 15 convs of layers 2-4 -> avg-pool -> mean over T -> head) plus,
for N>1, the RCCL all-gather of the logits.  Rank 0 prints ONE JSON line.  `roofline` prices the conv
kernels (all of their launches in a step, as scheduled) against the dense bf16 MFMA peak; `cpu_baseline`
is the PyTorch-CPU-eager oracle timed on this host's cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2516.6            # 256 CU x 4096 bf16 FLOP/clk/CU x 2.4 GHz, dense
HBM_PEAK_GBPS = 8000.0               # HBM3E spec
R18_CONV_MAC_PER_FRAME = 1_813_561_344   # conv MACs only, unpadded, 224x224


def conv_flops_per_frame(model):
    """Algorithmic conv FLOPs per frame from the loaded weights (cross-check of the constant)."""
    from video_asl_recognition_amd.spec import resnet_blocks
    S = model.spec.size
    mac = (S // 2) ** 2 * 64 * 147
    hw = S // 4
    for prefix, cin, planes, stride, ds in resnet_blocks("resnet18"):
        hw //= stride
        mac += hw * hw * planes * cin * 9 + hw * hw * planes * planes * 9
        if ds:
            mac += hw * hw * planes * cin
    return 2 * mac


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=20, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("ASL_CHUNK", "640")),
                    help="frames per backbone launch: 640 = 10 clips fills whole rounds of workgroups in every layer")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("ASL_STREAMS", "2")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="launcher / rendezvous / reporting path only: gloo on CPU with a stand-in step "
                         "(no kernels; the JSON line says so and is not a measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.cpu_dry_run:
        return dry_run(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from video_asl_recognition_amd import AslClipModel, gather_logits, init_state_dict
    from video_asl_recognition_amd.spec import SPECS

    spec = SPECS["resnet18_mean"]
    B, T, S = args.clips, args.frames, spec.size
    model = AslClipModel(spec, device=str(dev), frames_per_chunk=args.chunk, streams=args.streams)
    # the product path takes its weights from its own initialiser on every rank; the oracle is built (with the same
    # values) only for the cpu_baseline leg, after the timed region
    use_oracle = rank == 0 and world == 1 and not args.no_cpu_baseline
    state = init_state_dict(spec, seed=0)
    model.load_state_dict(state)

    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    clips = torch.randn(B, T, 3, S, S, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)

    def step():
        return gather_logits(model(clips))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert out.shape == (B * world, spec.num_classes)
    clips_per_s = world * B * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: every gather-GEMM (conv) launch of one step, timed
        # with events on the launch stream in an instrumented pass over the same inputs
        conv_ms, launches = time_conv_launches(model, clips, passes=3)
        flops_step = conv_flops_per_frame(model) * B * T
        assert conv_flops_per_frame(model) == 2 * R18_CONV_MAC_PER_FRAME
        step_s = elapsed / args.steps
        # `frac` prices the schedule that was timed: the step's conv FLOPs over the step's wall time (chunks
        # overlap on `streams` HIP streams, so per-launch durations cannot simply be added up); the
        # event-bracketed serial pass (each conv launch alone on one stream) is kept as serial_*.
        achieved = flops_step / step_s / 1e12
        serial = flops_step / (conv_ms * 1e-3) / 1e12
        # counters cannot be read inside the timed run: they come from the newest committed summary of this same
        # command, and are reported only when that summary was collected on these kernel sources with this chunking
        ctr = profiled_counters()
        traffic = hbm_gbps = mfma_busy = None
        profiled = {"source": None, "counters_stale": None}
        if ctr:
            from video_asl_recognition_amd._lib import source_hash
            nstreams = max(1, min(args.streams, -(-B * T // args.chunk)))
            run_chunk = min(args.chunk, -(-B * T // nstreams))          # frames per launch in this run (model._run_backbone)
            same_cfg = ctr.get("frames_per_chunk") == run_chunk
            stale = ctr.get("source_hash") != source_hash()
            profiled = {"source": ctr["file"], "collected_on_source_hash": ctr.get("source_hash"),
                        "this_source_hash": source_hash(), "counters_stale": bool(stale),
                        "frames_per_chunk": ctr.get("frames_per_chunk"), "same_chunking": bool(same_cfg),
                        "note": ctr["note"]}
            if same_cfg and not stale:
                step_bytes = ctr["hbm_bytes_per_chunk"] * (B * T / float(ctr["frames_per_chunk"]))
                traffic = round(step_bytes / launches)
                hbm_gbps = round(step_bytes / step_s / 1e9, 1)
                mfma_busy = ctr.get("mfma_busy")
        try:
            pre = time_with_preprocess(model, B, T, args.steps, args.warmup, dev)
        except Exception as e:          # the secondary figure must never cost the headline line
            pre = {"value": None, "unit": "clips/s", "error": f"{type(e).__name__}: {e}"[:200]}
        result = {
            "metric": "clips/sec end-to-end (T=64, 224x224, bf16)",
            "value": round(clips_per_s, 2),
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "resnet18_mean: ResNet-18 per frame + mean over T + Linear(512,100), "
                                   f"T={T}, {S}x{S}, {B} clips/GPU/step, random-init weights, clips resident in HBM",
                       "clips_per_gpu": B, "frames_per_clip": T, "frames_per_chunk": args.chunk, "streams": args.streams,
                       "parallelism": f"batch-sharded x{world}, all-gather of logits"},
            "roofline": {
                "bound": "mfma",
                "kernel": "the conv MFMA kernels of one step (stem_pool, block_c64_pair, conv3x3_pp / conv3x3_wide, "
                          "gather_gemm), as scheduled in the timed region",
                "achieved": round(achieved, 2),
                "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "profiled": profiled,
                "hbm_gbps": hbm_gbps,
                "hbm_peak_gbps": HBM_PEAK_GBPS,
                "hbm_frac": None if hbm_gbps is None else round(hbm_gbps / HBM_PEAK_GBPS, 4),
                "mfma_busy": mfma_busy,
                "launches_per_step": launches,
                "algorithmic_gflop_per_step": round(flops_step / 1e9, 1),
                "serial_achieved": round(serial, 2),
                "serial_frac": round(serial / MFMA_PEAK_TFLOPS, 4),
                "serial_avg_launch_us": round(conv_ms * 1e3 / launches, 2),
            },
        }
        result["with_preprocess"] = pre
        if use_oracle:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import asl_oracle as O
            oracle = O.build("resnet18_mean", seed=0)
            oracle.load_state_dict(state, strict=False)
            result["cpu_baseline"], result["parity"] = cpu_baseline(oracle, model, T, S)
        # the line goes out before the process group is torn down: a hang or error in the teardown cannot lose it
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script (one per GPU) under torch.distributed.run and return their exit
    code.  Runs before anything in this process has touched a GPU; the ranks are child processes.  The ranks'
    stdout is passed through untouched (rank 0's JSON line); their stderr is passed through too and also kept
    per rank, so that when the job fails the last lines of every rank's stderr are printed once more, rank by rank."""
    import glob
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory(prefix="asl_bench_ranks_") as logdir:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "--tee", "2", "--log-dir", logdir,
               os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.run(cmd, env=env).returncode
        if rc != 0:
            print(f"bench.py: the {n} ranks exited with code {rc}; last lines of each rank's stderr:", file=sys.stderr)
            for path in sorted(glob.glob(os.path.join(logdir, "**", "stderr.log"), recursive=True)):
                try:
                    with open(path, errors="replace") as f:
                        tail = f.read().splitlines()[-25:]
                except OSError:
                    continue
                rank_dir = os.path.basename(os.path.dirname(path))
                print(f"--- rank {rank_dir} ({len(tail)} lines) ---", file=sys.stderr)
                for ln in tail:
                    print("  " + ln, file=sys.stderr)
            sys.stderr.flush()
    return rc


def dry_run(args, world, rank):
    """The multi-rank plumbing of this script on CPU (gloo): same fences, same max-over-ranks timing,
    same single JSON line from rank 0; the step is a stand-in matmul, so `value` measures nothing."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    B, Cn = args.clips, 100
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, 64, generator=g)
    w = torch.randn(64, Cn, generator=torch.Generator().manual_seed(0))

    def step():
        y = x @ w
        if world == 1:
            return y
        out = torch.empty(world * B, Cn)
        dist.all_gather_into_tensor(out, y)
        return out

    def fence():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert out.shape == (B * world, Cn)
    if rank == 0:
        print(json.dumps({"metric": "clips/sec end-to-end (T=64, 224x224, bf16)", "value": None, "unit": "clips/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "CPU DRY RUN of the launcher and reporting path: stand-in step, "
                                                 "no kernels, not a measurement",
                                     "gathered_rows": int(out.shape[0])}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def profiled_counters():
    """Counter figures of the newest committed rocprofv3 PMC summary of this same command
    (profiles/r*_counters.json, written by tools/pmc_summary.py from separate --pmc passes): HBM bytes
    per chunk of frames with the gfx950 FETCH_SIZE correction applied, and the MFMA-busy ratio.
    Counters cannot be read from inside the timed run, so these are the last profiled values; None
    when no summary is present."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
    if not paths:
        return None
    try:
        with open(paths[-1]) as f:
            d = json.load(f)
        d["file"] = os.path.basename(paths[-1])
        d["note"] = (f"HBM bytes per conv launch = profiled (2*FETCH_SIZE + WRITE_SIZE) KiB of all kernels of a "
                     f"{d.get('frames_per_chunk', '?')}-frame chunk x chunks per step / conv launches "
                     f"({os.path.basename(paths[-1])})")
        return d
    except (OSError, ValueError, KeyError):
        return None


def time_with_preprocess(model, B, T, steps, warmup, dev, h0=240, w0=320):
    """The second figure: the same step fed with uint8 frames (h0 x w0, resident in HBM) instead of normalised clips --
    bilinear resize + normalise + pack in one kernel, then the same backbone and head, then softmax / top-5
    (classify_clip).  Timed after the headline region, same fences, this rank only."""
    g = torch.Generator(device=dev).manual_seed(7)
    frames = torch.randint(0, 256, (B, T, h0, w0, 3), generator=g, device=dev, dtype=torch.uint8)
    for _ in range(max(1, warmup)):
        model.classify_clip(frames, topk=5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx, _ = model.classify_clip(frames, topk=5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert idx.shape == (B, 5)
    return {"value": round(B / dt, 2), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 3), "scope": "this rank",
            "input": f"uint8 frames [{B},{T},{h0},{w0},3] resident in HBM -> resize + normalise -> backbone -> "
                     f"mean over T -> head -> softmax / top-5"}


def time_conv_launches(model, clips, passes=3):
    """Sum of the conv launch durations of one step (ms) and their count: every op of the plan issued
    alone on the current stream, conv launches bracketed by events on that stream."""
    import ctypes as C
    from video_asl_recognition_amd import _lib as L
    B, T, _, S, _ = clips.shape
    n_total = B * T
    plan = model._get_plans(min(n_total, model.frames_per_chunk), 1)[0]
    sarr = (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
    one = (L.Op * 1)()
    esz = clips.element_size()
    best = None
    for _ in range(passes):
        total_ms, launches = 0.0, 0
        for c0 in range(0, n_total, plan["nmax"]):
            n = min(plan["nmax"], n_total - c0)
            one[0] = L.make_op(L.OP_PACK_CLIPS, None, [clips.data_ptr() + c0 * 3 * S * S * esz, plan["x0"].data_ptr()],
                               [int(clips.dtype == torch.float32), n, S, plan["in_halo"]])
            L.run_ops(one, 1, sarr, 1, "pack")
            evs = []
            for rec in plan["ops"]:
                if rec["params"] is not None:
                    rec["params"].n = n
                one[0] = rec["op"]
                if rec["n_slot"] is not None:
                    one[0].i[rec["n_slot"]] = n
                if rec["conv"]:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    L.run_ops(one, 1, sarr, 1, "conv")
                    e1.record()
                    evs.append((e0, e1))
                else:
                    L.run_ops(one, 1, sarr, 1, "op")
            torch.cuda.synchronize()
            total_ms += sum(a.elapsed_time(b) for a, b in evs)
            launches += len(evs)
        best = total_ms if best is None else min(best, total_ms)
    return best, launches


def cpu_baseline(oracle, model, T, S):
    """The PyTorch-CPU-eager oracle on this host: 1 clip (T frames), 1 warm-up + 3 timed forwards;
    the same clip through the GPU path gives the parity figures."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asl_oracle as O
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable      # the job's CPU share, when the box sets one
    # thread counts tried: the job's share, and (when more CPUs are visible) twice that; best one is reported
    tries = sorted({max(1, min(share, usable)), max(1, min(2 * share, usable))})
    clip = O.make_clips(1, T, S, seed=0)
    reps, best = 3, None
    with torch.no_grad():
        for threads in tries:
            torch.set_num_threads(threads)
            ref = oracle(clip)
            t0 = time.perf_counter()
            for _ in range(reps):
                ref = oracle(clip)
            dt = (time.perf_counter() - t0) / reps
            if best is None or dt < best[0]:
                best = (dt, threads)
    dt, threads = best
    y = model(clip.cuda()).cpu()
    rel = float((y - ref).norm() / ref.norm())
    base = {"value": round(1.0 / dt, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "host_logical_cpus": logical, "host_physical_cores": physical_cores(), "usable_cpus": usable,
            "threads_tried": tries,
            "sample": f"1 clip x {T} frames {S}x{S} fp32, torch.nn CPU eager oracle, 1 warm-up + {reps} timed forwards "
                      f"per thread count, best reported"}
    parity = {"rel_l2_vs_cpu_oracle": rel, "argmax_identical": bool(torch.equal(y.argmax(1), ref.argmax(1))),
              "tolerance": 1e-3}
    return base, parity


def physical_cores():
    """Physical cores of the host from /proc/cpuinfo (distinct (physical id, core id) pairs), or None."""
    try:
        cores, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cores.add((phys, line.split(":")[1].strip()))
        return len(cores) or None
    except OSError:
        return None


if __name__ == "__main__":
    main()
