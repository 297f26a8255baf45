"""CPU: host-side logic -- weight packing layouts, spec/oracle parameter agreement, sharding."""
import numpy as np
import pytest
import torch

import asl_oracle as O
from video_asl_recognition_amd.packing import bn_scale_bias, pack_conv, pack_linear, pack_stem
from video_asl_recognition_amd.spec import SPECS, param_shapes
from video_asl_recognition_amd.dist import shard_batch


@pytest.mark.parametrize("name", list(SPECS))
def test_param_shapes_match_oracle_state_dict(name):
    m = O.build(name)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert sd == param_shapes(SPECS[name])
    assert SPECS[name].num_classes == O.SPECS[name].num_classes
    assert (SPECS[name].frames, SPECS[name].size, SPECS[name].batch) == (
        O.SPECS[name].frames, O.SPECS[name].size, O.SPECS[name].batch)


def test_pack_conv_layout():
    w = torch.randn(10, 40, 3, 3).bfloat16().float()
    pc = pack_conv(w, "cpu")
    assert pc.w.shape == (128, 9 * 64) and pc.cin_phys == 64 and pc.w_rows == 128 and pc.k == (1, 3, 3)
    p = pc.w.float().view(128, 3, 3, 64)
    for o, c, r, q in [(0, 0, 0, 0), (9, 39, 2, 1), (3, 17, 1, 2)]:
        assert p[o, r, q, c] == w[o, c, r, q]
    assert p[:, :, :, 40:].abs().sum() == 0 and p[10:].abs().sum() == 0


def test_pack_conv3d_layout():
    w = torch.randn(6, 32, 3, 3, 3).bfloat16().float()
    pc = pack_conv(w, "cpu")
    p = pc.w.float().view(128, 3, 3, 3, 32)
    assert p[5, 2, 0, 1, 31] == w[5, 31, 2, 0, 1] and pc.k == (3, 3, 3)


def test_pack_stem_layout():
    w = torch.randn(64, 3, 7, 7).bfloat16().float()
    pc = pack_stem(w, "cpu")
    assert pc.w.shape == (128, 224)
    p = pc.w.float().view(128, 7, 8, 4)
    assert p[11, 4, 6, 2] == w[11, 2, 4, 6]
    assert p[:, :, 7].abs().sum() == 0 and p[:, :, :, 3].abs().sum() == 0


def test_pack_linear_pads_k():
    w = torch.randn(100, 42).bfloat16().float()
    pc = pack_linear(w, "cpu", bias=torch.zeros(100))
    assert pc.w.shape == (128, 64) and torch.equal(pc.w[:100, :42].float(), w) and pc.w[:, 42:].abs().sum() == 0


def test_bn_scale_bias():
    sd = {"bn.weight": torch.tensor([2.0]), "bn.bias": torch.tensor([0.5]), "bn.running_mean": torch.tensor([1.0]),
          "bn.running_var": torch.tensor([4.0])}
    s, b = bn_scale_bias(sd, "bn", 0.0, "cpu")
    assert torch.allclose(s, torch.tensor([1.0])) and torch.allclose(b, torch.tensor([-0.5]))


def test_shard_batch():
    assert [shard_batch(16, r, 8) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
    with pytest.raises(ValueError):
        shard_batch(8, 4, 4)
