"""Host-side steps around the clip -> logits path (checkpoint import, sampler, keypoints, label map).
The reference holds no code or fixtures for them (/root/reference/README.md:1): the contracts are the ones
stated in video_asl_recognition_amd/clipio.py, checked here against independent numpy restatements and
properties.  Parity unpinned by the reference."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_asl_recognition_amd import SPECS, param_shapes  # noqa: E402
from video_asl_recognition_amd import clipio  # noqa: E402


def _fake_sd(spec, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(shp, generator=g) for k, shp in param_shapes(spec).items()}


@pytest.mark.parametrize("wrap", ["plain", "state_dict", "module", "torchvision", "nested"])
def test_checkpoint_import_name_mapping(tmp_path, wrap):
    spec = SPECS["resnet18_mean"]
    sd = _fake_sd(spec)
    obj = dict(sd)
    obj["backbone.bn1.num_batches_tracked"] = torch.tensor(7)
    if wrap == "state_dict":
        obj = {"epoch": 3, "state_dict": obj}
    elif wrap == "module":
        obj = {"model": {"module." + k: v for k, v in obj.items()}}
    elif wrap == "torchvision":
        obj = {(k[len("backbone."):] if k.startswith("backbone.") else k): v for k, v in obj.items()}
    elif wrap == "nested":
        obj = {"model_state_dict": {"model." + k: v for k, v in obj.items()}, "optimizer": {"lr": 0.1}}
    path = str(tmp_path / "ckpt.pth")
    torch.save(obj, path)
    got = clipio.load_checkpoint(path, "resnet18_mean")
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k


def test_checkpoint_without_tensors_is_rejected(tmp_path):
    path = str(tmp_path / "bad.pt")
    torch.save({"epoch": 1, "note": "nothing here"}, path)
    with pytest.raises(ValueError):
        clipio.load_checkpoint(path, "tiny")


def test_sampler_uniform_matches_restatement():
    for n in (1, 5, 31, 32, 33, 64, 100, 257, 1000):
        for T in (1, 8, 32, 64):
            got = clipio.sample_frame_indices(n, T)
            if n <= T:
                want = [min(i, n - 1) for i in range(T)]
            else:   # centre of each of T equal segments, float restatement
                want = [min(int(np.floor((i + 0.5) * n / T + 1e-9)), n - 1) for i in range(T)]
            assert got == want, (n, T)
            assert len(got) == T and all(0 <= a <= b < n for a, b in zip(got, got[1:] + [n - 1]))
    assert clipio.sample_frame_indices(64, 64) == list(range(64))
    assert clipio.sample_frame_indices(100, 4, "head") == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        clipio.sample_frame_indices(0, 4)
    with pytest.raises(ValueError):
        clipio.sample_frame_indices(10, 4, "random")


def test_center_crop_and_sample_clip():
    assert clipio.center_crop_box(240, 320) == (0, 40, 240, 240)
    assert clipio.center_crop_box(320, 240) == (40, 0, 240, 240)
    assert clipio.center_crop_box(100, 100) == (0, 0, 100, 100)
    v = torch.arange(10 * 6 * 8 * 3, dtype=torch.int64).reshape(10, 6, 8, 3).to(torch.uint8)
    c = clipio.sample_clip(v, 4)
    idx = clipio.sample_frame_indices(10, 4)
    assert c.shape == (4, 6, 6, 3) and c.is_contiguous()
    assert torch.equal(c, v[idx][:, :, 1:7])
    assert clipio.sample_clip(v[:2], 5, crop=False).shape == (5, 6, 8, 3)


def test_keypoint_normalisation_matches_numpy():
    rng = np.random.default_rng(0)
    kp = rng.uniform(0, 300, size=(2, 7, 21, 2)).astype(np.float32)
    kp[0, 1] = 0.0                       # undetected hand: all zeros
    kp[1, 2, 5, 0] = np.nan              # undetected hand: NaN
    t = torch.from_numpy(kp)
    img = clipio.normalize_keypoints(t, 240, 320, "image").numpy()
    want = kp / np.array([320.0, 240.0], dtype=np.float32)
    want[0, 1] = 0
    want[1, 2] = 0
    np.testing.assert_allclose(img, want, rtol=1e-6, atol=0)
    wr = clipio.normalize_keypoints(t, 240, 320, "wrist").numpy()
    rel = kp - kp[..., :1, :]
    sc = np.maximum(np.sqrt((rel ** 2).sum(-1)).max(-1), 1e-6)[..., None, None]
    w2 = rel / sc
    w2[0, 1] = 0
    w2[1, 2] = 0
    np.testing.assert_allclose(wr, w2, rtol=1e-5, atol=1e-6)
    assert np.isfinite(wr).all() and np.abs(wr[0, 0, 0]).max() == 0          # wrist at the origin
    # translation / scale invariance of the wrist form
    wr2 = clipio.normalize_keypoints(torch.from_numpy(kp[0, 0] * 3 + 11), 1, 1, "wrist").numpy()
    np.testing.assert_allclose(wr2, wr[0, 0], rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        clipio.normalize_keypoints(torch.zeros(3, 20, 2), 1, 1)


def test_label_map_forms_and_decode(tmp_path):
    forms = {
        "wlasl.json": [{"gloss": "book", "instances": []}, {"gloss": "drink"}, {"gloss": "computer"}],
        "list.json": ["book", "drink", "computer"],
        "idx2gloss.json": {"0": "book", "2": "computer", "1": "drink"},
        "gloss2idx.json": {"book": 0, "computer": 2, "drink": 1},
    }
    for name, obj in forms.items():
        p = tmp_path / name
        p.write_text(json.dumps(obj))
        lm = clipio.LabelMap.from_json(str(p))
        assert lm.glosses == ["book", "drink", "computer"], name
    lm2 = clipio.LabelMap.from_json(str(tmp_path / "list.json"), num_classes=2)
    assert len(lm2) == 2
    with pytest.raises(ValueError):
        clipio.LabelMap.from_json(str(tmp_path / "list.json"), num_classes=5)
    (tmp_path / "gap.json").write_text(json.dumps({"0": "a", "2": "b"}))
    with pytest.raises(ValueError):
        clipio.LabelMap.from_json(str(tmp_path / "gap.json"))
    out = lm.decode(torch.tensor([[2, 0], [1, 2]]), torch.tensor([[0.75, 0.25], [0.5, 0.125]]))
    assert out == [[("computer", 0.75), ("book", 0.25)], [("drink", 0.5), ("computer", 0.125)]]
