"""CPU: the C-ABI library loads, exports every symbol include/asl_hip.h declares, and rejects
bad geometry before touching a GPU (validation-only calls: nothing is launched)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "asl_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    from video_asl_recognition_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in asl_hip.h but not exported by libaslhip.so"
    assert set(_lib.EXPORTS) == set(names), "ctypes prototypes and header disagree"


def test_version_and_strerror():
    from video_asl_recognition_amd import _lib
    assert _lib.lib.asl_version() >= 100
    assert "invalid" in _lib.strerror(_lib.ASL_EINVAL)
    assert _lib.strerror(0) == "ok"


def test_struct_sizes_match_header_layout():
    from video_asl_recognition_amd import _lib as L
    assert C.sizeof(L.Geom) == 24 and C.sizeof(L.BlockParams) == 9 * 4
    assert C.sizeof(L.Conv2dParams) == 21 * 4 and C.sizeof(L.EntryParams) == 9 * 4
    assert C.sizeof(L.Op) == 8 + 8 + 10 * 8 + 8 * 4 + 6 * 4
    assert C.sizeof(L.StemParams) == 13 * 4
    assert C.sizeof(L.Conv3dParams) == 22 * 4
    assert C.sizeof(L.LinearParams) == 10 * 4
    assert C.sizeof(L.LstmParams) == 7 * 4 and C.sizeof(L.AttnParams) == 6 * 4
    assert C.sizeof(L.Pool3dParams) == 18 * 4
    assert C.sizeof(L.GemmParams) == 12 * 4 + 3 * 24 + 4 * 4 + 3 * 8 + 6 * 4


def test_bad_geometry_is_rejected_without_launch():
    from video_asl_recognition_amd import _lib as L
    fake = C.c_void_p(0x1000)       # 16-byte aligned, never dereferenced: validation fails first
    p = L.Conv2dParams(n=1, h=8, w=8, cin=48, in_halo=1, cout=64, kh=3, kw=3, stride=1, pad=1, out_halo=1,
                       out_ctot=64, out_coff=0, relu=1, w_rows=128)
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL  # cin % 32
    p.cin = 64
    p.in_halo = 0                                                                                            # halo < pad
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL
    p.in_halo = 1
    p.w_rows = 64                                                                                            # rows % 128
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL
    p.w_rows = 128
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), None, fake, None, None, None, fake, None) == L.ASL_EINVAL  # null x
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), C.c_void_p(0x1004), fake, None, None, None, fake, None) == L.ASL_EINVAL
    lp = L.LinearParams(m=4, n=10, k=40, ldx=40, ldy=10, ldres=0, relu=0, out_f32=1, res_f32=0, w_rows=128)
    assert L.lib.asl_linear_bf16(C.byref(lp), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL      # k % 32
    sp = L.StemParams(n=1, h=32, w=32, in_halo=2, cout=64, kh=7, kw=7, stride=2, pad=3, out_halo=1, relu=1, w_rows=128)
    assert L.lib.asl_stem_conv_bf16(C.byref(sp), fake, fake, None, None, fake, None) == L.ASL_EINVAL          # halo < pad
    assert L.lib.asl_pack_clips(fake, 1, 1, 31, 3, fake, None) == L.ASL_EINVAL                                # odd size
    assert L.lib.asl_softmax_topk(fake, 1, 10, 17, fake, fake, None) == L.ASL_EINVAL                          # k > 16
    assert L.lib.asl_avgpool_nhwc(fake, 1, 7, 7, 24 * 8, 1, fake, None, 0, 0, None) == L.ASL_EINVAL              # c/8 not pow2
    ep = L.EntryParams(n=1, h=7, w=8, cin=64, cout=128, out_halo=1, ds_halo=0, relu=1, w_rows=128)
    assert L.lib.asl_block_entry_s2_bf16(C.byref(ep), fake, fake, None, None, None, None, None, fake, None, None) == L.ASL_EINVAL  # odd h
    p.out_s2d = 1                                                                                            # cells need an even grid ...
    p.h = 7
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL
    p.h, p.out_halo = 8, 0                                                                                   # ... and halo 1
    assert L.lib.asl_conv2d_nhwc_bf16(C.byref(p), fake, fake, None, None, None, fake, None) == L.ASL_EINVAL
    with pytest.raises(RuntimeError, match="invalid argument"):
        L.check(L.ASL_EINVAL, "probe")


def test_static_lstm_rejects_requests_beyond_its_built_in_state():
    """the allocation-free LSTM entry point refuses 6*batch*hidden floats above its array before any HIP call"""
    from video_asl_recognition_amd import _lib as L
    fake = C.c_void_p(0x1000)
    p = L.LstmParams(batch=129, t=4, hidden=512, ld_gx=4096, ld_y=2048, split_y=1, ysum_scale=0.25)
    assert 6 * 129 * 512 > L.LSTM_STATIC_STATE_FLOATS
    assert L.lib.asl_lstm_bidir_layer_static(C.byref(p), fake, fake, fake, None, None) == L.ASL_ETOOBIG
    assert "capacity" in L.strerror(L.ASL_ETOOBIG)
    p.batch = 0
    assert L.lib.asl_lstm_bidir_layer_static(C.byref(p), fake, fake, fake, None, None) == L.ASL_EINVAL
    hdr = open(HEADER).read()
    assert re.search(r"#define\s+ASL_LSTM_STATIC_STATE_FLOATS\s+\(6 \* 64 \* 1024\)", hdr)


def test_model_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from video_asl_recognition_amd import AslClipModel
    with pytest.raises(RuntimeError, match="no CPU path"):
        AslClipModel("tiny")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "video_asl_recognition_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "asl_oracle" not in src and "np_restatement" not in src, f


def test_one_hip_runtime_in_the_process():
    """Loading the library before torch used to leave two HIP runtimes in the process (the system one for these
    kernels, the torch wheel's for torch): kernel launches then failed with 'no ROCm-capable device'.  Whatever is
    imported first, exactly one libamdhip64 may be mapped."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from video_asl_recognition_amd import _lib\n"
            "import torch\n"
            "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
            "print(len(paths), sorted(paths))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout
