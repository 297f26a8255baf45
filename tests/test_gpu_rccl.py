"""GPU: the RCCL calls of the multi-GPU path, executed for real at world size 1.

A one-GPU box cannot host two RCCL ranks, so the N > 1 bench never runs here; what can run is every collective call it
makes, with the same arguments, on a one-rank "nccl" process group: library load, communicator creation bound to the
device, barrier with device_ids, all_gather_into_tensor of a logits block, all_reduce(MAX) of a float64 scalar, teardown.
In a child process, so that the test session itself never holds a process group."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["ASL_ROOT"])
    from video_asl_recognition_amd import AslClipModel, gather_logits
    from video_asl_recognition_amd.dist import shard_batch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    dist.barrier(device_ids=[0])
    m = AslClipModel("tiny", seed=0)
    clips = torch.randn(4, 8, 3, 112, 112, device=dev)
    lo, hi = shard_batch(4, 0, 1)
    local = m(clips[lo:hi])
    assert gather_logits(local) is local                     # one rank: nothing to gather
    out = torch.empty_like(local)
    dist.all_gather_into_tensor(out, local.contiguous())     # the call gather_logits makes for world > 1
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    print("rccl world-1 ok")
""")


@pytest.mark.gpu
def test_rccl_calls_of_the_bench_at_world_size_one(gpu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", ASL_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
