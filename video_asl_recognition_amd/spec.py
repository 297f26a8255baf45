"""Model specifications of the clip -> logits path: one per BASELINE.json config.

The reference snapshot ships no model code (/root/reference/README.md:1 is its only
content), so these specs are the contract: architecture, the shapes each config is quoted
at, and the parameter names (torchvision-style for the ResNets, torch.nn names for the
LSTM / transformer) that ``AslClipModel.load_state_dict`` accepts.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple


@dataclass(frozen=True)
class ModelSpec:
    name: str
    backbone: str       # "tiny" | "resnet18" | "resnet50" | "i3d"
    temporal: str       # "mean" | "bilstm" | "transformer" | "i3d" (pooled inside the backbone)
    num_classes: int
    frames: int         # T the config is quoted at
    size: int           # S: square input resolution
    batch: int          # B the config is quoted at
    keypoints: bool = False
    feat_dim: int = 512
    lstm_hidden: int = 512
    lstm_layers: int = 2
    d_model: int = 512
    n_heads: int = 8
    ffn_dim: int = 2048
    n_layers: int = 4
    kp_dim: int = 42
    kp_hidden: int = 128


SPECS: Dict[str, ModelSpec] = {
    "tiny": ModelSpec("tiny", "tiny", "mean", 100, 8, 112, 2, feat_dim=64),
    "resnet18_mean": ModelSpec("resnet18_mean", "resnet18", "mean", 100, 32, 224, 8, feat_dim=512),
    "i3d": ModelSpec("i3d", "i3d", "i3d", 100, 64, 224, 16, feat_dim=1024),
    "resnet50_bilstm": ModelSpec("resnet50_bilstm", "resnet50", "bilstm", 100, 64, 224, 16, feat_dim=2048),
    "twostream_transformer": ModelSpec("twostream_transformer", "resnet18", "transformer", 2000, 128, 224, 8,
                                       keypoints=True, feat_dim=512),
}

RESNET_LAYERS = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet50": ("bottleneck", (3, 4, 6, 3))}

I3D_MIXED = (
    ("mixed_3b", (64, 96, 128, 16, 32, 32)), ("mixed_3c", (128, 128, 192, 32, 96, 64)),
    ("pool_4a", None),
    ("mixed_4b", (192, 96, 208, 16, 48, 64)), ("mixed_4c", (160, 112, 224, 24, 64, 64)),
    ("mixed_4d", (128, 128, 256, 24, 64, 64)), ("mixed_4e", (112, 144, 288, 32, 64, 64)),
    ("mixed_4f", (256, 160, 320, 32, 128, 128)),
    ("pool_5a", None),
    ("mixed_5b", (256, 160, 320, 32, 128, 128)), ("mixed_5c", (384, 192, 384, 48, 128, 128)),
)


def _bn(shapes, prefix, c):
    shapes[prefix + ".weight"] = (c,)
    shapes[prefix + ".bias"] = (c,)
    shapes[prefix + ".running_mean"] = (c,)
    shapes[prefix + ".running_var"] = (c,)


def resnet_blocks(kind: str):
    """Yield (prefix, cin, planes, stride, has_downsample) for every residual block."""
    block, layers = RESNET_LAYERS[kind]
    exp = 1 if block == "basic" else 4
    cin = 64
    for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
        for j in range(n):
            stride = 2 if (i > 0 and j == 0) else 1
            ds = stride != 1 or cin != planes * exp
            yield f"backbone.layer{i + 1}.{j}", cin, planes, stride, ds
            cin = planes * exp


def param_shapes(spec: ModelSpec) -> Dict[str, Tuple[int, ...]]:
    """Name -> shape of every tensor a state_dict for `spec` must hold."""
    s: Dict[str, Tuple[int, ...]] = {}
    if spec.backbone == "tiny":
        s["backbone.conv1.weight"] = (32, 3, 3, 3)
        s["backbone.conv1.bias"] = (32,)
        s["backbone.conv2.weight"] = (64, 32, 3, 3)
        s["backbone.conv2.bias"] = (64,)
    elif spec.backbone in RESNET_LAYERS:
        block = RESNET_LAYERS[spec.backbone][0]
        s["backbone.conv1.weight"] = (64, 3, 7, 7)
        _bn(s, "backbone.bn1", 64)
        for prefix, cin, planes, stride, ds in resnet_blocks(spec.backbone):
            if block == "basic":
                s[prefix + ".conv1.weight"] = (planes, cin, 3, 3)
                _bn(s, prefix + ".bn1", planes)
                s[prefix + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(s, prefix + ".bn2", planes)
                cout = planes
            else:
                s[prefix + ".conv1.weight"] = (planes, cin, 1, 1)
                _bn(s, prefix + ".bn1", planes)
                s[prefix + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(s, prefix + ".bn2", planes)
                s[prefix + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                _bn(s, prefix + ".bn3", planes * 4)
                cout = planes * 4
            if ds:
                s[prefix + ".downsample.0.weight"] = (cout, cin, 1, 1)
                _bn(s, prefix + ".downsample.1", cout)
    elif spec.backbone == "i3d":
        def unit(prefix, cin, cout, k):
            s[prefix + ".conv.weight"] = (cout, cin, k, k, k)
            _bn(s, prefix + ".bn", cout)
        unit("backbone.conv_1a", 3, 64, 7)
        unit("backbone.conv_2b", 64, 64, 1)
        unit("backbone.conv_2c", 64, 192, 3)
        cin = 192
        for name, w in I3D_MIXED:
            if w is None:
                continue
            p = "backbone." + name
            unit(p + ".b0", cin, w[0], 1)
            unit(p + ".b1a", cin, w[1], 1)
            unit(p + ".b1b", w[1], w[2], 3)
            unit(p + ".b2a", cin, w[3], 1)
            unit(p + ".b2b", w[3], w[4], 3)
            unit(p + ".b3b", cin, w[5], 1)
            cin = w[0] + w[2] + w[4] + w[5]
        s["backbone.logits.weight"] = (spec.num_classes, cin, 1, 1, 1)
        s["backbone.logits.bias"] = (spec.num_classes,)
        return s
    else:
        raise ValueError(spec.backbone)

    f = spec.feat_dim
    if spec.temporal == "mean":
        s["fc.weight"] = (spec.num_classes, f)
        s["fc.bias"] = (spec.num_classes,)
    elif spec.temporal == "bilstm":
        h = spec.lstm_hidden
        for layer in range(spec.lstm_layers):
            i = f if layer == 0 else 2 * h
            for sfx in ("", "_reverse"):
                s[f"lstm.weight_ih_l{layer}{sfx}"] = (4 * h, i)
                s[f"lstm.weight_hh_l{layer}{sfx}"] = (4 * h, h)
                s[f"lstm.bias_ih_l{layer}{sfx}"] = (4 * h,)
                s[f"lstm.bias_hh_l{layer}{sfx}"] = (4 * h,)
        s["fc.weight"] = (spec.num_classes, 2 * h)
        s["fc.bias"] = (spec.num_classes,)
    elif spec.temporal == "transformer":
        d = spec.d_model
        s["kp_mlp.0.weight"] = (spec.kp_hidden, spec.kp_dim)
        s["kp_mlp.0.bias"] = (spec.kp_hidden,)
        s["kp_mlp.2.weight"] = (spec.kp_hidden, spec.kp_hidden)
        s["kp_mlp.2.bias"] = (spec.kp_hidden,)
        s["fuse.weight"] = (d, f + spec.kp_hidden)
        s["fuse.bias"] = (d,)
        for l in range(spec.n_layers):
            p = f"encoder.layers.{l}"
            s[p + ".self_attn.in_proj_weight"] = (3 * d, d)
            s[p + ".self_attn.in_proj_bias"] = (3 * d,)
            s[p + ".self_attn.out_proj.weight"] = (d, d)
            s[p + ".self_attn.out_proj.bias"] = (d,)
            s[p + ".linear1.weight"] = (spec.ffn_dim, d)
            s[p + ".linear1.bias"] = (spec.ffn_dim,)
            s[p + ".linear2.weight"] = (d, spec.ffn_dim)
            s[p + ".linear2.bias"] = (d,)
            s[p + ".norm1.weight"] = (d,)
            s[p + ".norm1.bias"] = (d,)
            s[p + ".norm2.weight"] = (d,)
            s[p + ".norm2.bias"] = (d,)
        s["fc.weight"] = (spec.num_classes, d)
        s["fc.bias"] = (spec.num_classes,)
    return s
