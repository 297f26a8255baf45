"""GPU: end-to-end parity of AslClipModel against the CPU oracle (identical seeded random-init
weights and synthetic clips).  Bar: ||y - y_ref||_2 / ||y_ref||_2 <= 1e-3 over a batch's logits
and arg-max identical per clip, plus agreement with the committed golden logits."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import asl_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-3


def rel_l2(y, ref):
    return float((y - ref).norm() / ref.norm())


def _model(name, seed=0, **kw):
    from video_asl_recognition_amd import AslClipModel
    om = O.build(name, seed=seed)
    m = AslClipModel(name, **kw)
    m.load_state_dict(om.state_dict())
    return om, m


def test_tiny_forward_matches_oracle_and_golden(gpu):
    om, m = _model("tiny")
    g = np.load(os.path.join(GOLDEN, "tiny_fwd_b2t8s112.npz"))
    clips = O.make_clips(2, 8, 112, 0)
    y = m(clips.cuda()).cpu()
    with torch.no_grad():
        ref = om(clips)
    assert rel_l2(y, ref) <= REL_TOL
    assert rel_l2(y, torch.from_numpy(g["logits"])) <= REL_TOL
    assert torch.equal(y.argmax(1), ref.argmax(1))


@pytest.mark.parametrize("B,T,chunk", [(2, 8, 256), (2, 8, 5), (1, 32, 16)])
def test_resnet18_forward_matches_oracle(gpu, B, T, chunk):
    om, m = _model("resnet18_mean", frames_per_chunk=chunk)
    clips = O.make_clips(B, T, 224, 0)
    y = m(clips.cuda()).cpu()
    with torch.no_grad():
        ref = om(clips)
    assert y.shape == (B, 100) and y.dtype == torch.float32
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))
    if (B, T) == (2, 8):
        g = np.load(os.path.join(GOLDEN, "resnet18_mean_fwd_b2t8s224.npz"))
        assert rel_l2(y, torch.from_numpy(g["logits"])) <= REL_TOL


def test_resnet18_cell_entry_path_matches_oracle(gpu):
    """the opt-in one-launch stride-2 block entries (cell tensors written by the previous block's kernel): same bar as
    the default path, and within bf16 rounding of it"""
    om, m = _model("resnet18_mean", frames_per_chunk=8, s2d_entry=True)
    _, m0 = _model("resnet18_mean", frames_per_chunk=8, s2d_entry=False)
    clips = O.make_clips(2, 8, 224, 0)
    y, y0 = m(clips.cuda()).cpu(), m0(clips.cuda()).cpu()
    with torch.no_grad():
        ref = om(clips)
    kinds = [int(r["op"].kind) for r in m._plan[0]["ops"]]
    assert kinds.count(14) == 3, "three stride-2 entries expected on the one-launch path"      # ASL_OP_BLOCK_ENTRY
    assert rel_l2(y, ref) <= REL_TOL and torch.equal(y.argmax(1), ref.argmax(1))
    assert rel_l2(y, y0) <= 3e-4


@pytest.mark.parametrize("name,B,T,size", [("resnet18_mean", 2, 8, 224), ("resnet50_bilstm", 2, 6, 224)])
def test_parity_mode_matches_oracle_more_closely(gpu, name, B, T, size):
    """parity mode (hi + lo activation pairs through every convolution): same oracle, an error several times below
    the default path's"""
    om, m = _model(name, parity_mode=True)
    _, m0 = _model(name)
    clips = O.make_clips(B, T, size, 0)
    y, y0 = m(clips.cuda()).cpu(), m0(clips.cuda()).cpu()
    with torch.no_grad():
        ref = om(clips)
    kinds = [int(r["op"].kind) for r in m._plan[0]["ops"]]
    assert 15 in kinds and 13 not in kinds       # ASL_OP_MAXPOOL_SPLIT present, no fused ASL_OP_BASIC_BLOCK
    e, e0 = rel_l2(y, ref), rel_l2(y0, ref)
    assert e <= REL_TOL and torch.equal(y.argmax(1), ref.argmax(1))
    assert e <= 0.5 * e0 + 2e-5, (e, e0)


def test_resnet18_bf16_input_equals_fp32_input(gpu):
    """inputs are bf16-representable, so handing them over as bf16 must change nothing"""
    _, m = _model("resnet18_mean")
    clips = O.make_clips(1, 4, 224, 3).cuda()
    assert torch.equal(m(clips), m(clips.to(torch.bfloat16)))


def test_resnet18_frame_features(gpu):
    om, m = _model("resnet18_mean")
    clips = O.make_clips(1, 6, 224, 2)
    f, fb = m.frame_features(clips.cuda(), want_bf16=True)
    with torch.no_grad():
        ref = om.frame_features(clips)[0]
    assert rel_l2(f.cpu(), ref) <= 5e-3          # per-frame features: no averaging over T yet
    assert torch.equal(fb.float(), f.to(torch.bfloat16).float())


def test_classify_clip_matches_oracle_and_golden(gpu):
    om, m = _model("resnet18_mean")
    g = np.load(os.path.join(GOLDEN, "resnet18_mean_cls_b1t4s224.npz"))
    frames = O.make_frames(1, 4, 120, 160, 0)
    idx, prob = m.classify_clip(frames[0].cuda(), topk=5)       # [T,H,W,3] form
    ridx, rprob = O.classify_clip(om, frames, topk=5)
    assert torch.equal(idx.cpu(), ridx) and np.array_equal(idx.cpu().numpy(), g["idx"])
    torch.testing.assert_close(prob.cpu(), rprob, rtol=2e-2, atol=1e-5)
    om2, m2 = _model("tiny")
    frames = O.make_frames(2, 8, 60, 80, 0)
    idx, prob = m2.classify_clip(frames.cuda(), topk=5)
    ridx, rprob = O.classify_clip(om2, frames, topk=5, size=112)
    assert torch.equal(idx.cpu()[:, 0], ridx[:, 0])
    torch.testing.assert_close(prob.cpu().sort(1).values, rprob.sort(1).values, rtol=2e-3, atol=1e-6)


def test_preprocess_surface(gpu):
    _, m = _model("tiny")
    frames = O.make_frames(1, 2, 60, 80, 1)
    torch.testing.assert_close(m.preprocess(frames.cuda()).cpu(), O.preprocess(frames, 112), rtol=1e-5, atol=2e-5)


def test_errors_are_loud(gpu):
    from video_asl_recognition_amd import AslClipModel
    m = AslClipModel("resnet18_mean")
    with pytest.raises(RuntimeError, match="load_state_dict"):
        m(torch.zeros(1, 1, 3, 224, 224))
    m = AslClipModel("resnet18_mean", seed=0)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 3, 112, 112))
    with pytest.raises(KeyError):
        m.load_state_dict({"fc.weight": torch.zeros(100, 512)})
    with pytest.raises(ValueError):
        m.classify_clip(torch.zeros(2, 8, 8, 3))


def test_full_size_linearity_property(gpu):
    """Size-independent check at the metric's full shape (T=64, 224^2): with every BatchNorm bias and
    conv-free path unchanged, the mean-pool head is linear in the per-frame features, so logits of a clip
    equal the head applied to the mean of its frame features -- checked on device at full size without
    the (slow) oracle."""
    _, m = _model("resnet18_mean", frames_per_chunk=128)
    clips = O.make_clips(2, 64, 224, 9).cuda()
    y = m(clips)
    f, _ = m.frame_features(clips)
    ref = f.view(2, 64, -1).mean(1) @ m._fc_w.T + m._fc_b
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    # permuting the frames of a clip must not change a mean-pooled clip's logits beyond fp32 order effects
    perm = torch.randperm(64)
    torch.testing.assert_close(m(clips[:, perm]), y, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name,B,T,sub", [("i3d", 16, 64, 3), ("resnet50_bilstm", 16, 64, 5), ("twostream_transformer", 8, 128, 3)])
def test_full_batch_logits_do_not_depend_on_the_batch(gpu, name, B, T, sub):
    """Size-independent check at the configurations' full batch and clip length (too slow for the CPU oracle):
    a clip's logits are the same whether it travels in the full batch or in ragged sub-batches of `sub` clips --
    the clip axis of the 3-D backbone, the batch rows of the recurrence's MFMA groups (16 at a time), the attention
    workgroups and every GEMM's row tiling must all keep clips apart."""
    from video_asl_recognition_amd import AslClipModel
    m = AslClipModel(name, seed=5)
    clips = O.make_clips(B, T, m.spec.size, 21).cuda()
    kp = O.make_keypoints(B, T, 21).cuda() if m.spec.keypoints else None
    y = m(clips, kp).clone()
    assert y.shape == (B, m.spec.num_classes) and bool(torch.isfinite(y).all())
    parts = [m(clips[i:i + sub], None if kp is None else kp[i:i + sub]).clone() for i in range(0, B, sub)]
    torch.testing.assert_close(torch.cat(parts), y, rtol=1e-5, atol=1e-5)
    assert float((y[0] - y[1]).abs().max()) > 0                                  # clips do get different logits


def test_chunking_and_streams_do_not_change_results(gpu):
    """Ragged chunking over two streams (300 frames as 128 + 128 + 44), one stream, and one big chunk give the
    same per-frame features bit for bit: every output pixel is computed by the same K-ordered sum whatever tile,
    workgroup or stream it lands in."""
    from video_asl_recognition_amd import AslClipModel
    clips = O.make_clips(3, 100, 224, 4).cuda()
    ref = None
    for chunk, streams in ((512, 1), (128, 2), (128, 1), (77, 3)):
        m = AslClipModel("resnet18_mean", frames_per_chunk=chunk, streams=streams, seed=3)
        f, _ = m.frame_features(clips)
        f = f.clone()
        if ref is None:
            ref = f
        else:
            assert torch.equal(f, ref), f"chunk={chunk} streams={streams} changed the features"
        del m


@pytest.mark.parametrize("name,B,T", [("resnet18_mean", 3, 8), ("resnet50_bilstm", 2, 16), ("i3d", 2, 16),
                                      ("twostream_transformer", 2, 32)])
def test_forward_replays_from_a_hip_graph(gpu, name, B, T):
    """A forward captured with torch.cuda.graph (chunk streams, branch streams and their events, the LSTM's per-step
    launches and memsets all inside the capture) replays on new input contents and gives what a direct forward gives."""
    from video_asl_recognition_amd import AslClipModel
    m = AslClipModel(name, seed=6)
    S = m.spec.size
    static = O.make_clips(B, T, S, 1).cuda().to(torch.bfloat16)
    kp = O.make_keypoints(B, T, 2).cuda() if m.spec.keypoints else None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                               # warm-up off the default stream: plans, streams, LDS sizes
        for _ in range(2):
            m(static, kp)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(static, kp)
    for seed in (3, 4):
        fresh = O.make_clips(B, T, S, seed).cuda().to(torch.bfloat16)
        static.copy_(fresh)
        g.replay()
        torch.cuda.synchronize()
        got = out.clone()
        ref = m(fresh, kp)
        assert bool(torch.isfinite(got).all()) and torch.equal(got, ref)
    # the same through AslClipModel.capture
    cap = m.capture(B, T)
    fresh = O.make_clips(B, T, S, 7).cuda()
    assert torch.equal(cap(fresh, kp).clone(), m(fresh, kp))
    with pytest.raises(ValueError, match="captured for clips"):
        cap(fresh[:1], kp)


def test_classify_clip_resizes_into_the_clip_tensor_the_stem_reads(gpu, monkeypatch):
    """uint8 frames are resized and normalised straight into bf16 clips [n][3][S][S], which the stem reads in place:
    the logits are those of forward(preprocess(frames)) bit for bit (ragged chunks, two streams, two calls), and the
    packed-input road (ASL_PLANAR_STEM=0) agrees within the stem kernels' accumulation order."""
    from video_asl_recognition_amd import AslClipModel
    frames = O.make_frames(2, 4, 60, 80, 3).cuda()
    m = AslClipModel("resnet18_mean", seed=3, frames_per_chunk=3, streams=2)
    assert m._planar_input()
    logits = m(m.preprocess(frames)).clone()
    rp, ri = F.softmax(logits, -1).topk(5, -1)
    for _ in range(2):
        idx, prob = m.classify_clip(frames, topk=5)
        assert torch.equal(idx, ri)
        torch.testing.assert_close(prob, rp, rtol=1e-5, atol=1e-8)
    monkeypatch.setenv("ASL_PLANAR_STEM", "0")
    m2 = AslClipModel("resnet18_mean", seed=3, frames_per_chunk=3, streams=2)
    assert not m2._planar_input()
    idx2, prob2 = m2.classify_clip(frames, topk=5)
    assert torch.equal(idx2[:, 0], ri[:, 0])
    torch.testing.assert_close(prob2.sort(1).values, rp.sort(1).values, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("name,B,T", [("resnet50_bilstm", 5, 16), ("twostream_transformer", 3, 32)])
def test_heads_per_chunk_do_not_change_results(gpu, name, B, T):
    """The temporal encoder + head run chunk by chunk on the chunks' streams (opt-in; chunks of one and of two clips,
    a ragged last chunk) against one head over the whole batch after the backbone: same logits, forward and
    classify_clip."""
    from video_asl_recognition_amd import AslClipModel
    clips = O.make_clips(B, T, 224, 12).cuda()
    kp = O.make_keypoints(B, T, 13).cuda() if name.startswith("twostream") else None
    m0 = AslClipModel(name, seed=4, chunk_heads=False)
    ref = m0(clips, kp).clone()
    frames = O.make_frames(B, T, 60, 80, 3).cuda()
    ridx, rprob = m0.classify_clip(frames, kp, topk=3)
    assert bool(torch.isfinite(ref).all()) and float((ref[0] - ref[1]).abs().max()) > 0
    for chunk, streams in ((T, 2), (2 * T, 2), (2 * T, 1), (T + 5, 3)):
        m = AslClipModel(name, seed=4, frames_per_chunk=chunk, streams=streams, chunk_heads=True)
        assert m._heads_per_chunk()
        for _ in range(2):                                  # the second call reuses plans, streams and freed buffers
            torch.testing.assert_close(m(clips, kp), ref, rtol=1e-5, atol=1e-5)
        idx, prob = m.classify_clip(frames, kp, topk=3)
        assert torch.equal(idx, ridx)
        torch.testing.assert_close(prob, rprob, rtol=1e-4, atol=1e-6)
        del m


def test_i3d_streams_and_chunk_pairs_do_not_change_results(gpu):
    """The 3-D backbone on one stream, with its Inception branches on three streams, and with chunks paired side by
    side on six streams (one-clip chunks: clips 0 and 1 together, clip 2 alone) gives the same logits bit for bit."""
    from video_asl_recognition_amd import AslClipModel
    clips = O.make_clips(3, 16, 224, 8).cuda()
    ref = None
    for chunk, streams in ((1024, 1), (1024, 2), (16, 2), (32, 2)):
        m = AslClipModel("i3d", frames_per_chunk=chunk, streams=streams, seed=2)
        y = m(clips).clone()
        y2 = m(clips).clone()                       # workspaces and events are reused by the second call
        assert torch.equal(y, y2)
        if ref is None:
            ref = y
            assert bool(torch.isfinite(y).all()) and float((y[0] - y[1]).abs().max()) > 0
        else:
            assert torch.equal(y, ref), f"chunk={chunk} streams={streams} changed the logits"
        del m


def test_fused_basic_blocks_give_the_same_logits(gpu, monkeypatch):
    """layer 1 run as one kernel per BasicBlock (the default) against the two-launch form and the oracle"""
    om, m1 = _model("resnet18_mean")
    monkeypatch.setenv("ASL_FUSE_BLOCKS", "0")
    _, m0 = _model("resnet18_mean")
    assert m1.fuse_blocks and not m0.fuse_blocks
    clips = O.make_clips(2, 8, 224, 3)
    y0, y1 = m0(clips.cuda()).cpu(), m1(clips.cuda()).cpu()
    ref = om(clips)
    assert rel_l2(y1, ref) <= REL_TOL and torch.equal(y1.argmax(1), ref.argmax(1))
    assert rel_l2(y1, y0) <= 5e-4       # different summation order inside the MFMAs: a few bf16 roundings flip


def test_resnet50_fused_exits_give_the_same_logits(gpu, monkeypatch):
    """ResNet-50 with a bottleneck's last 1x1 conv and the next block's first one in one launch (56 x 56 stage) gives, bit
    for bit, the logits of the plan that launches them apart."""
    from video_asl_recognition_amd import AslClipModel
    from video_asl_recognition_amd import _lib as L
    clips = O.make_clips(2, 8, 224, 5).cuda()
    m = AslClipModel("resnet50_bilstm", seed=9)
    kinds = [int(r["op"].kind) for r in m._get_plans(16, 1)[0]["ops"]]
    assert kinds.count(L.OP_CONV1X1_CHAIN) == 2 and kinds.count(L.OP_CONV1X1_DUAL) == 4
    y = m(clips).clone()
    monkeypatch.setenv("ASL_CHAIN_1X1", "0")
    m1 = AslClipModel("resnet50_bilstm", seed=9)
    assert L.OP_CONV1X1_CHAIN not in [int(r["op"].kind) for r in m1._get_plans(16, 1)[0]["ops"]]
    assert torch.equal(m1(clips), y)


def test_single_frame_single_clip(gpu):
    """smallest clip the surface accepts: B=1, T=1"""
    om, m = _model("resnet18_mean")
    clips = O.make_clips(1, 1, 224, 11)
    y = m(clips.cuda()).cpu()
    ref = om(clips)
    assert torch.linalg.norm(y - ref) / torch.linalg.norm(ref) < 1e-3
    assert int(y.argmax()) == int(ref.argmax())


def test_resnet50_bilstm_matches_oracle(gpu):
    om, m = _model("resnet50_bilstm", frames_per_chunk=16)
    g = np.load(os.path.join(GOLDEN, "resnet50_bilstm_fwd_b2t8s224.npz"))
    clips = O.make_clips(2, 8, 224, 0)
    y = m(clips.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))


def test_resnet50_bilstm_full_length_matches_golden(gpu):
    """the configuration's own sequence length: 64 frames through ResNet-50 and the 2 x 64-step recurrence"""
    om, m = _model("resnet50_bilstm", frames_per_chunk=64)
    g = np.load(os.path.join(GOLDEN, "resnet50_bilstm_fwd_b1t64s224.npz"))
    clips = O.make_clips(1, 64, 224, 0)
    y = m(clips.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))


def test_twostream_transformer_full_length_matches_golden(gpu):
    """128 tokens (the attention kernel's limit), 2000 classes"""
    om, m = _model("twostream_transformer", frames_per_chunk=128)
    g = np.load(os.path.join(GOLDEN, "twostream_transformer_fwd_b1t128s224.npz"))
    clips, kp = O.make_clips(1, 128, 224, 0), O.make_keypoints(1, 128, 0)
    y = m(clips.cuda(), kp.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert y.shape == (1, 2000)
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))


def test_i3d_two_clips_full_length_matches_golden(gpu):
    """two 64-frame clips: the clip axis of the 3-D backbone (separate zero-padded depth ranges per clip)"""
    om, m = _model("i3d", frames_per_chunk=128)
    g = np.load(os.path.join(GOLDEN, "i3d_fwd_b2t64s224.npz"))
    clips = O.make_clips(2, 64, 224, 0)
    y = m(clips.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert y.shape == (2, 100)
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))


def test_twostream_transformer_matches_oracle(gpu):
    om, m = _model("twostream_transformer", frames_per_chunk=32)
    g = np.load(os.path.join(GOLDEN, "twostream_transformer_fwd_b2t16s224.npz"))
    clips, kp = O.make_clips(2, 16, 224, 0), O.make_keypoints(2, 16, 0)
    y = m(clips.cuda(), kp.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert y.shape == (2, 2000)
    assert rel_l2(y, ref) <= REL_TOL, rel_l2(y, ref)
    assert torch.equal(y.argmax(1), ref.argmax(1))
    with pytest.raises(ValueError):
        m(clips.cuda())


def _i3d_case(T):
    om, m = _model("i3d", frames_per_chunk=64)
    g = np.load(os.path.join(GOLDEN, f"i3d_fwd_b1t{T}s224.npz"))
    clips = O.make_clips(1, T, 224, 0)
    y = m(clips.cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    assert y.shape == (1, 100)
    assert torch.equal(y.argmax(1), ref.argmax(1))
    return rel_l2(y, ref)


def test_i3d_matches_oracle(gpu):
    """The configuration's own clip length, T = 64: logits within 1e-3 of the fp32 oracle, arg-max identical."""
    rel = _i3d_case(64)
    assert rel <= REL_TOL, rel


I3D_SHORT_CLIP_TOL = 2e-3


def test_i3d_short_clip_out_of_config_T16(gpu):
    """OUT OF CONFIG: a 16-frame clip is not a length any configuration names (the I3D configuration is T = 64).  It
    leaves 2x7x7 positions for the final average instead of 8x7x7, so the per-activation bf16 rounding averages down
    about half as much and the logits land at 1.4e-3 of the fp32 oracle -- outside the 1e-3 bar, which is stated
    at T = 64.  Kept as a regression guard at 2e-3 with the arg-max still required identical."""
    rel = _i3d_case(16)
    assert rel <= I3D_SHORT_CLIP_TOL, rel


def test_checkpoint_import_then_forward_matches_oracle(gpu, tmp_path):
    """a torchvision-named, DataParallel-wrapped *.pth round-trips through load_checkpoint into the HIP model"""
    from video_asl_recognition_amd import AslClipModel, clipio
    om = O.build("resnet18_mean", seed=5)
    sd = om.state_dict()
    ck = {"epoch": 9, "state_dict": {"module." + (k[len("backbone."):] if k.startswith("backbone.") else k): v
                                     for k, v in sd.items()}}
    path = str(tmp_path / "r18.pth")
    torch.save(ck, path)
    m = AslClipModel("resnet18_mean")
    m.load_state_dict(clipio.load_checkpoint(path, "resnet18_mean"))
    clips = O.make_clips(1, 4, 224, 2)
    y = m(clips.cuda()).cpu()
    with torch.no_grad():
        ref = om(clips)
    assert rel_l2(y, ref) <= REL_TOL
    assert torch.equal(y.argmax(1), ref.argmax(1))


def test_clip_stream_equals_direct_classification(gpu):
    """double-buffered host->device streaming returns exactly what classify_clip returns per batch,
    including a ragged last batch and buffer reuse across three batches"""
    from video_asl_recognition_amd import clipio
    _, m = _model("resnet18_mean")
    batches = [O.make_frames(2, 4, 60, 80, seed=s) for s in (0, 1)] + [O.make_frames(1, 4, 60, 80, seed=2)]
    direct = [m.classify_clip(b.cuda(), topk=3) for b in batches]
    torch.cuda.synchronize()
    streamed = list(clipio.ClipStream(m, topk=3)(batches))
    torch.cuda.synchronize()
    assert len(streamed) == 3
    for (i0, p0), (i1, p1) in zip(direct, streamed):
        assert torch.equal(i0, i1) and torch.equal(p0, p1)
    pinned = list(clipio.ClipStream(m, topk=3)([b.pin_memory() for b in batches]))     # transferred without staging
    torch.cuda.synchronize()
    for (i0, p0), (i1, p1) in zip(direct, pinned):
        assert torch.equal(i0, i1) and torch.equal(p0, p1)
    assert list(clipio.ClipStream(m)([])) == []


def test_clip_stream_and_label_map_against_the_oracle(gpu, tmp_path):
    """streamed host batches -> top-k -> gloss strings, checked against the CPU oracle's classify_clip (not against
    the HIP path itself): indices identical, probabilities within 1e-3, decoded glosses those of the oracle's indices"""
    import json
    from video_asl_recognition_amd import clipio
    om, m = _model("resnet18_mean")
    batches = [O.make_frames(2, 4, 60, 80, seed=s) for s in (20, 21)] + [O.make_frames(1, 4, 60, 80, seed=22)]
    path = tmp_path / "labels.json"
    path.write_text(json.dumps([{"gloss": f"sign_{i:03d}", "instances": []} for i in range(100)]))
    labels = clipio.LabelMap.from_json(str(path), num_classes=100)
    for batch, (idx, prob) in zip(batches, clipio.ClipStream(m, topk=3)(batches)):
        ri, rp = O.classify_clip(om, batch, topk=3, size=224)
        assert torch.equal(idx.cpu(), ri)
        torch.testing.assert_close(prob.cpu(), rp, rtol=1e-3, atol=1e-5)
        decoded = labels.decode(idx, prob)
        assert [[g for g, _ in row] for row in decoded] == [[f"sign_{i:03d}" for i in row] for row in ri.tolist()]
        assert all(abs(p - q) < 1e-3 for row, ref in zip(decoded, rp.tolist()) for (_, p), q in zip(row, ref))


def test_clip_stream_survives_shape_changes_while_batches_are_in_flight(gpu):
    """batch shapes change in both directions mid-stream (buffers regrown while the previous batch's
    kernels are still queued): every result still equals the direct classification of that batch"""
    from video_asl_recognition_amd import clipio
    _, m = _model("resnet18_mean")
    shapes = [(3, 16, 60, 80), (1, 4, 48, 64), (4, 16, 72, 96), (1, 2, 60, 80), (3, 16, 60, 80), (2, 4, 48, 64)]
    batches = [O.make_frames(b, t, h, w, seed=10 + i) for i, (b, t, h, w) in enumerate(shapes)]
    direct = [m.classify_clip(b.cuda(), topk=3) for b in batches]
    torch.cuda.synchronize()
    cs = clipio.ClipStream(m, topk=3)
    for _ in range(2):                                   # second pass reuses the grown buffers
        streamed = list(cs(batches))
        torch.cuda.synchronize()
        assert len(streamed) == len(batches)
        for k, ((i0, p0), (i1, p1)) in enumerate(zip(direct, streamed)):
            assert torch.equal(i0, i1) and torch.equal(p0, p1), f"batch {k} {shapes[k]}"


def test_sampled_video_and_keypoints_feed_the_two_stream_model(gpu):
    """sampler + keypoint normalisation produce tensors classify_clip accepts; result equals the oracle's
    on the same prepared inputs"""
    from video_asl_recognition_amd import clipio
    om, m = _model("twostream_transformer")
    video = O.make_frames(1, 21, 72, 96, seed=4)[0]
    clip = clipio.sample_clip(video, 16)
    assert clip.shape == (16, 72, 72, 3)
    g = torch.Generator().manual_seed(0)
    kp_px = torch.rand(21, 21, 2, generator=g) * torch.tensor([96.0, 72.0])
    kp_px[3] = 0                                   # an undetected hand
    idx = clipio.sample_frame_indices(21, 16)
    kp = clipio.normalize_keypoints(kp_px[idx], 72, 96).bfloat16().float()   # both sides hold identical values
    i1, p1 = m.classify_clip(clip.cuda(), kp.cuda(), topk=5)
    i0, p0 = O.classify_clip(om, clip[None], kp[None], topk=5, size=224)
    assert torch.equal(i1.cpu()[:, 0], i0[:, 0])
    assert float((p1.cpu() - p0).abs().max()) <= 1e-3
