"""CPU, world_size 2 (gloo): the multi-GPU layer's semantics -- contiguous batch shards and an
all-gather of per-rank logits that equals the single-process result, rank order preserved."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_asl_recognition_amd.dist import gather_logits, shard_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, Cn = (6 if world == 2 else 24), 10
        full = torch.arange(B * Cn, dtype=torch.float32).view(B, Cn) * 0.5   # "logits" of the whole batch
        lo, hi = shard_batch(B, rank, world)
        got = gather_logits(full[lo:hi].clone())
        torch.save({"ok": bool(torch.equal(got, full)), "shape": tuple(got.shape)}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_allgather_logits_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert res["ok"] and res["shape"] == (6, 10)


def test_gather_is_identity_without_process_group():
    x = torch.randn(3, 4)
    assert gather_logits(x) is x


def test_bench_launches_its_own_ranks_for_gpus_2():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start the two ranks itself
    (one JSON line from rank 0, exit code of the ranks); --cpu-dry-run swaps the kernels for a stand-in so
    the launcher, rendezvous, fences and reporting run here without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--clips", "8", "--cpu-dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["gathered_rows"] == 16 and "DRY RUN" in d["config"]["workload"] and d["value"] is None
    # ranks that fail (zero timed steps leave nothing to report) make the launcher's exit code non-zero, and the
    # launcher repeats the end of every rank's stderr, rank by rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "0", "--warmup", "0",
                        "--cpu-dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--- rank 0" in r.stderr and "--- rank 1" in r.stderr, r.stderr[-3000:]
    tail = r.stderr[r.stderr.index("--- rank 0"):]
    assert "Error" in tail, tail[-2000:]


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_bench_dry_run_world_4_and_8(world):
    """The launcher, rendezvous, fences, max-over-ranks timing and the single JSON line at the rank counts the
    scaling run uses (gloo on CPU, stand-in step)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--clips", "4", "--cpu-dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["config"]["gathered_rows"] == 4 * world


@pytest.mark.parametrize("world", [4, 8])
def test_allgather_logits_world_4_and_8(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert res["ok"] and res["shape"] == (24, 10)
