"""GPU: asl_lstm_bidir_layer_static (state in the library's built-in device array, no allocation)
against asl_lstm_bidir_layer (state in caller scratch) through the C ABI.  The two share every kernel
and the same launch order, so outputs must be bit-identical.  Two calls in a row, the second on a
smaller input, so stale contents of the shared array from the first call would show; then the
capacity limit.  The static entry point is also checked on its own against torch.nn.LSTM (fp32, CPU),
at a batch that needs three 16-row passes of the recurrence kernel."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _inputs(B, T, H, seed):
    g = torch.Generator().manual_seed(seed)
    gx = (torch.randn(B * T, 8 * H, generator=g) * 0.8).cuda()
    w_hh = (torch.randn(2, 4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(torch.bfloat16).cuda()
    return gx, w_hh


def _run(L, B, T, H, gx, w_hh, static):
    # outputs poisoned, so a skipped write shows
    y = torch.full((B * T, 4 * H), 7.0, dtype=torch.bfloat16, device="cuda")
    ysum = torch.full((B, 2 * H), 7.0, device="cuda")
    p = L.LstmParams(batch=B, t=T, hidden=H, ld_gx=8 * H, ld_y=4 * H, split_y=1, ysum_scale=1.0 / T)
    if static:
        L.check(L.lib.asl_lstm_bidir_layer_static(C.byref(p), _p(gx), _p(w_hh), _p(y), _p(ysum), None), "lstm static")
    else:
        state = torch.full((6 * B * H,), 3.0, device="cuda")
        L.check(L.lib.asl_lstm_bidir_layer(C.byref(p), _p(gx), _p(w_hh), _p(y), _p(ysum), _p(state), None), "lstm")
    torch.cuda.synchronize()
    return y.cpu(), ysum.cpu()


def test_static_state_lstm_equals_caller_state_lstm(gpu):
    from video_asl_recognition_amd import _lib as L
    # first call: the BiLSTM head's own shape with an odd batch (one partly filled 16-row pass); second: smaller
    # in every dimension, so it reuses a prefix of the array the first call left full of non-zero state
    for B, T, H, seed in ((5, 64, 512, 21), (2, 9, 32, 22)):
        gx, w_hh = _inputs(B, T, H, seed)
        y_ref, s_ref = _run(L, B, T, H, gx, w_hh, static=False)
        y, s = _run(L, B, T, H, gx, w_hh, static=True)
        assert torch.isfinite(s_ref).all() and float(s_ref.abs().max()) > 0
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), f"y differs at B={B},T={T},H={H}"
        assert torch.equal(s, s_ref), f"ysum differs at B={B},T={T},H={H}"


def test_static_state_lstm_rejects_oversize(gpu):
    from video_asl_recognition_amd import _lib as L
    B, T, H = 129, 1, 512                      # 6*129*512 floats: one batch row over the array's size
    assert 6 * B * H > L.LSTM_STATIC_STATE_FLOATS >= 6 * (B - 1) * H
    gx, w_hh = _inputs(B, T, H, 23)
    y = torch.full((B * T, 4 * H), 7.0, dtype=torch.bfloat16, device="cuda")
    p = L.LstmParams(batch=B, t=T, hidden=H, ld_gx=8 * H, ld_y=4 * H, split_y=1, ysum_scale=1.0)
    rc = L.lib.asl_lstm_bidir_layer_static(C.byref(p), _p(gx), _p(w_hh), _p(y), None, None)
    assert rc == L.ASL_ETOOBIG and "capacity" in L.strerror(rc)
    torch.cuda.synchronize()
    assert bool((y == 7.0).all()), "an oversize request must launch nothing"
    # the largest request that fits still runs and matches
    B = 128
    gx, w_hh = _inputs(B, T, H, 24)
    y_ref, _ = _run(L, B, T, H, gx, w_hh, static=False)
    y_st, _ = _run(L, B, T, H, gx, w_hh, static=True)
    assert torch.equal(y_st.view(torch.int16), y_ref.view(torch.int16))


def test_static_state_lstm_matches_nn_lstm(gpu):
    """The allocation-free entry point against torch.nn.LSTM itself (not only against its sibling): gx is the fp32
    input projection computed on the CPU, so the kernel under test is the recurrence alone; B = 33 takes three
    passes of 16 batch rows, the last one with a single live row."""
    from video_asl_recognition_amd import _lib as L
    B, T, I, H = 33, 10, 48, 64
    torch.manual_seed(31)
    lstm = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True, batch_first=True)
    with torch.no_grad():
        for q in lstm.parameters():
            q.copy_(q.to(torch.bfloat16).float())
        x = torch.randn(B, T, I)
        ref, _ = lstm(x)
        sd = dict(lstm.named_parameters())
        gx = torch.cat([x.reshape(B * T, I) @ sd["weight_ih_l0"].T + sd["bias_ih_l0"] + sd["bias_hh_l0"],
                        x.reshape(B * T, I) @ sd["weight_ih_l0_reverse"].T + sd["bias_ih_l0_reverse"] + sd["bias_hh_l0_reverse"]], 1)
        w_hh = torch.stack([sd["weight_hh_l0"], sd["weight_hh_l0_reverse"]], 0).to(torch.bfloat16).contiguous().cuda()
    y, ysum = _run(L, B, T, H, gx.contiguous().cuda(), w_hh, static=True)
    yy = y.float().view(B, T, 4 * H)
    # hi + lo halves of h carry ~16 mantissa bits; tolerance as in the kernel's own test against nn.LSTM
    torch.testing.assert_close(yy[..., :2 * H] + yy[..., 2 * H:], ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ysum, ref.mean(1), rtol=1e-4, atol=1e-5)


def test_static_state_lstm_replays_from_a_hip_graph(gpu):
    """The use the entry point exists for: a call captured into a HIP graph (nothing in it allocates; its state is zeroed
    by a kernel of the library) and replayed several times, each time on new contents of the same input buffer, against
    direct calls.  The first call on the device is made outside the capture (it looks the array's address up)."""
    from video_asl_recognition_amd import _lib as L
    B, T, H = 3, 12, 64
    gx0, w_hh = _inputs(B, T, H, 41)
    _run(L, B, T, H, gx0, w_hh, static=True)
    gx = gx0.clone()
    y = torch.full((B * T, 4 * H), 7.0, dtype=torch.bfloat16, device="cuda")
    ysum = torch.full((B, 2 * H), 7.0, device="cuda")
    p = L.LstmParams(batch=B, t=T, hidden=H, ld_gx=8 * H, ld_y=4 * H, split_y=1, ysum_scale=1.0 / T)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.asl_lstm_bidir_layer_static(C.byref(p), _p(gx), _p(w_hh), _p(y), _p(ysum), stream), "lstm static")
    for seed in (42, 43, 44):
        fresh, _ = _inputs(B, T, H, seed)
        gx.copy_(fresh)
        y.fill_(7.0)
        ysum.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        y_ref, s_ref = _run(L, B, T, H, fresh, w_hh, static=False)
        assert torch.equal(y.cpu().view(torch.int16), y_ref.view(torch.int16)), f"y differs on replay with seed {seed}"
        assert torch.equal(ysum.cpu(), s_ref), f"ysum differs on replay with seed {seed}"
