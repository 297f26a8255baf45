"""ctypes binding of libaslhip.so (the C ABI declared in include/asl_hip.h).

The library is built in-tree by ``video_asl_recognition_amd/csrc/Makefile`` (``__graft_entry__.build()``
runs it).  There is no fallback: if the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASL_LIB_PATH") or os.path.join(_HERE, "csrc", "libaslhip.so")   # override: diagnostic builds

ASL_OK, ASL_EINVAL, ASL_ENOTSUP, ASL_ETOOBIG = 0, -1, -2, -3
# kernel variants of asl_conv2d_nhwc_bf16 / asl_gather_gemm_bf16 (0 = the library chooses)
CONV_AUTO, CONV_GATHER, CONV_PATCH, CONV_PP, CONV_PP_BIG, CONV_WIDE, CONV_C64 = range(7)
LSTM_STATIC_STATE_FLOATS = 6 * 64 * 1024


class Geom(C.Structure):
    _fields_ = [("sn", C.c_int64), ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32), ("off0", C.c_int32)]


class GemmParams(C.Structure):
    _fields_ = [("nb", C.c_int32), ("od", C.c_int32), ("oh", C.c_int32), ("ow", C.c_int32),
                ("cout", C.c_int32), ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("ck", C.c_int32), ("tap_sd", C.c_int32), ("tap_sh", C.c_int32), ("tap_sw", C.c_int32),
                ("in_", Geom), ("out", Geom), ("res", Geom),
                ("relu", C.c_int32), ("out_f32", C.c_int32), ("res_f32", C.c_int32), ("w_rows", C.c_int32),
                ("x_elems", C.c_int64), ("y_elems", C.c_int64), ("res_elems", C.c_int64),
                ("variant", C.c_int32), ("out_s2d", C.c_int32),
                ("split_out", C.c_int32), ("out_lo", C.c_int32), ("res_lo", C.c_int32), ("reserved", C.c_int32)]


class Conv2dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "h", "w", "cin", "in_halo", "cout", "kh", "kw", "stride", "pad",
        "out_halo", "out_ctot", "out_coff", "res_halo", "res_ctot", "res_coff", "relu", "w_rows", "variant", "out_s2d", "out_split")]


class BlockParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "channels", "in_halo", "out_halo", "w_rows", "variant", "out_s2d")]


class EntryParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "cin", "cout", "out_halo", "ds_halo", "relu", "w_rows")]


class StemParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "h", "w", "in_halo", "cout", "kh", "kw", "stride", "pad", "out_halo", "relu", "w_rows", "out_split")]


class StemPoolParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "cout", "out_halo", "w_rows", "planar")]


class Conv3dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "d", "h", "w", "cin", "in_halo_d", "in_halo", "cout", "kd", "kh", "kw", "stride_d", "stride",
        "pad_d", "pad", "out_halo_d", "out_halo", "out_ctot", "out_coff", "relu", "w_rows", "variant")]


class Stem3dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "frames", "size", "cout", "out_halo_d", "out_halo", "relu", "w_rows")]


class Dual1x1Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "cin", "in_halo", "cin2", "in2_halo", "stride2", "cout", "out_halo",
                                        "out_ctot", "out_coff", "relu", "w_rows")]


class Chain1x1Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "cin", "cmid", "cout", "in_halo", "res_halo", "mid_halo", "out_halo",
                                        "w_rows_a", "w_rows_b")]


class LinearParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("m", "n", "k", "ldx", "ldy", "ldres", "relu", "out_f32", "res_f32", "w_rows")]


class Pool3dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "d", "h", "w", "c", "in_halo_d", "in_halo", "kd", "kh", "kw", "sd", "sh", "sw",
        "pd", "ph", "pw", "out_halo_d", "out_halo")]


class LstmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batch", "t", "hidden", "ld_gx", "ld_y", "split_y")] + [("ysum_scale", C.c_float)]


class Op(C.Structure):
    """asl_op: one launch as data, for asl_run_ops"""
    _fields_ = [("kind", C.c_int32), ("stream", C.c_int32), ("params", C.c_void_p), ("ptr", C.c_void_p * 10),
                ("i", C.c_int32 * 8), ("f", C.c_float * 6)]


OP_NOP = 0
(OP_CONV2D, OP_STEM_CONV, OP_STEM_POOL, OP_CONV3D, OP_LINEAR, OP_MAXPOOL3D, OP_MAXPOOL, OP_AVGPOOL, OP_PACK_CLIPS,
 OP_PREPROCESS, OP_MEAN_T, OP_HEAD, OP_BASIC_BLOCK, OP_BLOCK_ENTRY, OP_MAXPOOL_SPLIT, OP_CONV1X1_DUAL, OP_CONV1X1_CHAIN) = range(1, 18)


class AttnParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batch", "t", "heads", "head_dim", "ld_qkv", "ld_out")]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(or __graft_entry__.build()); this package has no non-HIP fallback")

# torch first: its wheel carries its own libamdhip64.so.7, and the HIP runtime this library binds to must be the one
# torch's streams and allocations live in.  Loaded the other way round, the process ends up with two HIP runtimes
# (the system one for these kernels, the wheel's for torch) and launches fail with "no ROCm-capable device".
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_P, _I, _F3 = C.c_void_p, C.c_int, C.POINTER(C.c_float)

_PROTOS = {
    "asl_version": ([], C.c_int),
    "asl_strerror": ([C.c_int], C.c_char_p),
    "asl_gather_gemm_bf16": ([C.POINTER(GemmParams), _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_conv2d_nhwc_bf16": ([C.POINTER(Conv2dParams), _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_basic_block_c64_bf16": ([C.POINTER(BlockParams)] + [_P] * 9, C.c_int),
    "asl_block_entry_s2_bf16": ([C.POINTER(EntryParams)] + [_P] * 10, C.c_int),
    "asl_stem_conv_bf16": ([C.POINTER(StemParams), _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_stem_pool_bf16": ([C.POINTER(StemPoolParams), _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_conv3d_ndhwc_bf16": ([C.POINTER(Conv3dParams), _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_stem3d_bf16": ([C.POINTER(Stem3dParams), _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_conv1x1_dual_bf16": ([C.POINTER(Dual1x1Params), _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_conv1x1_chain_bf16": ([C.POINTER(Chain1x1Params)] + [_P] * 11, C.c_int),
    "asl_linear_bf16": ([C.POINTER(LinearParams), _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_pack_clips": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    "asl_pack_clips_3d": ([_P, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "asl_preprocess_u8": ([_P, _I, _I, _I, _I, _I, _F3, _F3, _P, _P, _P], C.c_int),
    "asl_preprocess_u8_planar": ([_P, _I, _I, _I, _I, _F3, _F3, _P, _P], C.c_int),
    "asl_maxpool3x3s2_nhwc": ([_P, _I, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "asl_maxpool3x3s2_split_nhwc": ([_P, _I, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "asl_maxpool3d_ndhwc": ([C.POINTER(Pool3dParams), _P, _P, _P], C.c_int),
    "asl_avgpool_nhwc": ([_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P], C.c_int),
    "asl_mean_t_f32": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "asl_window_mean_t_f32": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    "asl_head_f32": ([_P, _I, _I, _P, _P, _I, _P, _P], C.c_int),
    "asl_softmax_topk": ([_P, _I, _I, _I, _P, _P, _P], C.c_int),
    "asl_lstm_bidir_layer": ([C.POINTER(LstmParams), _P, _P, _P, _P, _P, _P], C.c_int),
    "asl_lstm_bidir_layer_static": ([C.POINTER(LstmParams), _P, _P, _P, _P, _P], C.c_int),
    "asl_layernorm_f32": ([_P, _I, _I, _P, _P, C.c_float, _P, _I, _P, _P], C.c_int),
    "asl_add_rows_f32": ([_P, _I, _I, _P, _I, _P, _P, _I, _P], C.c_int),
    "asl_cast_pad_bf16": ([_P, _I, _I, _P, _I, _P], C.c_int),
    "asl_attention_fwd": ([C.POINTER(AttnParams), _P, _P, _P], C.c_int),
    "asl_run_ops": ([C.POINTER(Op), _I, C.POINTER(C.c_void_p), _I, C.POINTER(C.c_int)], C.c_int),
}

EXPORTS = tuple(_PROTOS)

for _name, (_args, _res) in _PROTOS.items():
    _fn = getattr(lib, _name)      # AttributeError here = the .so is stale; rebuild it
    _fn.argtypes = _args
    _fn.restype = _res


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, the
    Makefile and the public header), in name order.  Counter summaries are stamped with it so a summary collected on
    other kernels can be told apart from one collected on these."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files += [os.path.join(_HERE, "csrc", "Makefile"), os.path.join(os.path.dirname(_HERE), "include", "asl_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def strerror(rc: int) -> str:
    return lib.asl_strerror(rc).decode()


def make_op(kind, params=None, ptrs=(), ints=(), floats=(), stream=0) -> Op:
    o = Op()
    o.kind, o.stream = kind, stream
    o.params = C.addressof(params) if params is not None else None
    for k, p in enumerate(ptrs):
        o.ptr[k] = p
    for k, v in enumerate(ints):
        o.i[k] = v
    for k, v in enumerate(floats):
        o.f[k] = v
    return o


def run_ops(ops, count: int, streams, nstreams: int, what: str = "ops") -> None:
    """ops: ctypes array of Op; streams: ctypes array of c_void_p"""
    failed = C.c_int(-1)
    rc = lib.asl_run_ops(ops, count, streams, nstreams, C.byref(failed))
    if rc != 0:
        raise RuntimeError(f"{what}: op {failed.value} (kind {ops[failed.value].kind if failed.value >= 0 else '?'}) "
                           f"failed: rc={rc} ({strerror(rc)})")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"{what or 'asl call'} failed: rc={rc} ({strerror(rc)})")
