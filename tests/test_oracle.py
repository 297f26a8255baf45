"""CPU: the oracle's building blocks against the numpy restatement, and the committed golden
vectors against the oracle.  Parity is unpinned by the reference (it ships no vectors); these
tests pin the oracle to PyTorch's documented op semantics instead."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import asl_oracle as O
import np_restatement as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _np(t):
    return t.detach().numpy().astype(np.float64)


def test_conv_bn_relu_block_matches_numpy():
    torch.manual_seed(0)
    conv = nn.Conv2d(5, 7, 3, 2, 1, bias=False)
    bn = nn.BatchNorm2d(7).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
        x = torch.randn(2, 5, 9, 8)
        y = F.relu(bn(conv(x)))
    ref = np.maximum(R.batchnorm_eval(R.conv2d(_np(x), _np(conv.weight), None, 2, 1), _np(bn.weight), _np(bn.bias),
                                      _np(bn.running_mean), _np(bn.running_var)), 0)
    np.testing.assert_allclose(_np(y), ref, rtol=1e-5, atol=1e-5)


def test_bn_fold_identity():
    """conv + BN(eval) == conv * scale + bias with scale = g/sqrt(v+eps), bias = b - m*scale."""
    torch.manual_seed(1)
    conv = nn.Conv2d(4, 6, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(6).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
        x = torch.randn(1, 4, 6, 6)
        y = bn(conv(x))
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        bias = bn.bias - bn.running_mean * scale
        y2 = conv(x) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    assert torch.allclose(y, y2, rtol=1e-6, atol=1e-6)


def test_maxpool_matches_numpy():
    x = torch.randn(2, 3, 9, 7)
    np.testing.assert_array_equal(_np(F.max_pool2d(x, 3, 2, 1)), R.maxpool2d(_np(x)))


def test_conv3d_matches_numpy():
    torch.manual_seed(2)
    conv = nn.Conv3d(3, 4, 3, 2, 1, bias=False)
    x = torch.randn(1, 3, 5, 6, 7)
    np.testing.assert_allclose(_np(conv(x)), R.conv3d(_np(x), _np(conv.weight), 2, 1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("h0,w0,s", [(24, 32, 16), (10, 13, 16), (16, 16, 16)])
def test_bilinear_matches_numpy(h0, w0, s):
    x = torch.randint(0, 256, (1, 3, h0, w0)).float()
    y = F.interpolate(x, size=(s, s), mode="bilinear", align_corners=False, antialias=False)
    np.testing.assert_allclose(y.numpy(), R.bilinear_resize(x.numpy(), (s, s)), rtol=1e-5, atol=1e-3)


def test_lstm_gate_order_and_reverse_alignment():
    torch.manual_seed(3)
    lstm = nn.LSTM(6, 5, num_layers=1, bidirectional=True, batch_first=True)
    x = torch.randn(2, 7, 6)
    y, _ = lstm(x)
    params = {}
    for k, v in lstm.named_parameters():
        params[k.replace("_l0", "")] = _np(v)
    np.testing.assert_allclose(_np(y), R.lstm_bidir(_np(x), params, 5), rtol=1e-5, atol=1e-5)


def test_transformer_layer_matches_numpy():
    torch.manual_seed(4)
    torch.backends.mha.set_fastpath_enabled(False)
    layer = nn.TransformerEncoderLayer(16, 4, 32, dropout=0.0, activation="relu", batch_first=True).eval()
    x = torch.randn(2, 6, 16)
    with torch.no_grad():
        y = layer(x)
    p = dict(in_w=_np(layer.self_attn.in_proj_weight), in_b=_np(layer.self_attn.in_proj_bias),
             out_w=_np(layer.self_attn.out_proj.weight), out_b=_np(layer.self_attn.out_proj.bias),
             l1_w=_np(layer.linear1.weight), l1_b=_np(layer.linear1.bias), l2_w=_np(layer.linear2.weight),
             l2_b=_np(layer.linear2.bias), n1_w=_np(layer.norm1.weight), n1_b=_np(layer.norm1.bias),
             n2_w=_np(layer.norm2.weight), n2_b=_np(layer.norm2.bias))
    np.testing.assert_allclose(_np(y), R.encoder_layer_postln(_np(x), p, 4), rtol=1e-4, atol=1e-5)


def test_tiny_model_matches_numpy_end_to_end():
    m = O.build("tiny", seed=0)
    clips = O.make_clips(1, 2, 16, seed=0)
    with torch.no_grad():
        y = m(clips)
    sd = {k: _np(v) for k, v in m.state_dict().items()}
    np.testing.assert_allclose(_np(y), R.tiny_forward(_np(clips), sd), rtol=1e-4, atol=1e-5)


def test_params_and_inputs_are_bf16_representable():
    m = O.build("resnet18_mean", seed=0)
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.equal(v, v.to(torch.bfloat16).float()), k
    x = O.make_clips(1, 1, 8)
    assert torch.equal(x, x.to(torch.bfloat16).float())


def _sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("fname", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")) if os.path.isdir(GOLDEN) else [])
def test_golden_vectors_reproduce(fname):
    """Every committed fixture is regenerated from its recorded seeds by the oracle: same input
    bytes (sha256) and logits equal to fp32 round-off (thread count / ISA may reorder sums)."""
    g = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    name = str(g["model"])
    if int(g["heavy"]) and os.environ.get("ASL_HEAVY_GOLDEN", "0") != "1" and name in ("i3d", "resnet50_bilstm"):
        pytest.skip("heavy fixture (set ASL_HEAVY_GOLDEN=1)")
    B, T, S, seed = (int(g[k]) for k in ("B", "T", "S", "seed"))
    m = O.build(name, seed=seed)
    kind = str(g["kind"])
    kp = O.make_keypoints(B, T, seed) if name == "twostream_transformer" else None
    with torch.no_grad():
        if kind == "forward":
            x = O.make_clips(B, T, S, seed)
            assert _sha(x) == str(g["input_sha256"])
            y = m(x, kp)
            np.testing.assert_allclose(y.numpy(), g["logits"], rtol=2e-4, atol=2e-4)
        else:
            fr = O.make_frames(B, T, int(g["H0"]), int(g["W0"]), seed)
            assert _sha(fr) == str(g["input_sha256"])
            idx, prob = O.classify_clip(m, fr, kp, topk=int(g["topk"]), size=S)
            np.testing.assert_array_equal(idx.numpy(), g["idx"])
            np.testing.assert_allclose(prob.numpy(), g["prob"], rtol=1e-3, atol=1e-6)
