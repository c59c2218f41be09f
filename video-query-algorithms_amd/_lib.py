"""ctypes binding of libvqamd.so (the C ABI declared in include/vq_amd.h).

There is no CPU fallback: if the shared library is missing or fails to load the import raises,
and every call that fails inside the library raises ``VqError`` with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqamd.so")

ABI_VERSION = 11                 # include/vq_amd.h: VQ_ABI_VERSION
VQ_F32, VQ_F64 = 0, 1
VQ_LAYOUT_ROWS, VQ_LAYOUT_TILED = 0, 1
VQ_OP_CONV, VQ_OP_MAXPOOL, VQ_OP_AVGPOOL, VQ_OP_GLOBAL_AVGPOOL, VQ_OP_CONV_WINOGRAD, VQ_OP_CONV_WINOGRAD16 = 1, 2, 3, 4, 5, 6

_ERR_NAMES = {-1: "VQ_E_INVALID", -2: "VQ_E_HIP", -3: "VQ_E_NOMEM", -4: "VQ_E_STATE", -5: "VQ_E_UNSUPPORTED"}


class VqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_ERR_NAMES.get(code, "VQ_E_?"), code, msg))
        self.code = code


class LayerDesc(C.Structure):
    _fields_ = [("op", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("src_coff", C.c_int32),
                ("dst_coff", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32),
                ("stride", C.c_int32), ("pad", C.c_int32), ("relu", C.c_int32), ("ceil_mode", C.c_int32),
                ("has_bias", C.c_int32), ("seg_first", C.c_int32), ("seg_count", C.c_int32),
                ("pre_pool_k", C.c_int32), ("pre_pool_stride", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64)]


class ConvSegment(C.Structure):
    _fields_ = [("cout", C.c_int32), ("dst", C.c_int32), ("dst_coff", C.c_int32), ("relu", C.c_int32)]


class InputDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("s2d_pad", C.c_int32), ("s2d_kernel", C.c_int32), ("s2d_order", C.c_int32)]


class Tvl1Params(C.Structure):
    _fields_ = [("tau", C.c_float), ("lambda_", C.c_float), ("theta", C.c_float), ("epsilon", C.c_float), ("scale_step", C.c_float),
                ("nscales", C.c_int32), ("warps", C.c_int32), ("iterations", C.c_int32), ("bound", C.c_float)]


class TensorDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32)]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_I32, _I64, _U64, _F32, _F64 = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
_pI32, _pI64, _pF32, _pF64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double)

# name -> argtypes; every function returns int (status) unless listed in _SPECIAL
SIGNATURES = {
    "vq_device_count": [_pI32],
    "vq_format_feature_rows": [_P, _I64, _I32, _P, _I32, _P, _I64, _pI64],
    "vq_timer_create": [_PP], "vq_timer_start": [_P, _P], "vq_timer_stop": [_P, _P],
    "vq_timer_elapsed_ms": [_P, _pF32], "vq_timer_destroy": [_P],
    "vq_db_create": [_I64, _I32, _I32, _I32, _I32, _I32, _PP],
    "vq_db_destroy": [_P], "vq_db_set_stream": [_P, _P],
    "vq_db_shape": [_P, _pI64, _pI32, _pI32, _pI32, _pI32],
    "vq_db_set_layout": [_P, _I32], "vq_db_layout": [_P, _pI32],
    "vq_db_upload": [_P, _I64, _I64, _P], "vq_db_adopt_device": [_P, _P], "vq_db_set_present": [_P, _P],
    "vq_db_generate": [_P, _U64, _I64, _pF32], "vq_db_feats_devptr": [_P, _PP],
    "vq_db_set_query": [_P, _P], "vq_db_set_query_from_row": [_P, _I64, _P],
    "vq_bootstrap_targets": [_P, _I32, _I32, _P, _P, _I32, _F64, _I32, _P],
    "vq_db_bootstrap_target": [_P, _P, _I32, _P, _I32, _F64, _P, _I32],
    "vq_db_scan": [_P, _P, _I32], "vq_db_rescore": [_P, _P],
    "vq_db_scan_batch": [_P, _I32, _P, _P, _P], "vq_db_batch_scores_devptr": [_P, _PP, _pI32],
    "vq_db_read_similarities": [_P, _P, _P, _P], "vq_db_read_scores": [_P, _P],
    "vq_db_scores_devptr": [_P, _PP], "vq_db_avg_devptr": [_P, _PP], "vq_db_ne_devptr": [_P, _PP], "vq_db_read_rows": [_P, _P, _I32, _P], "vq_db_read_scores_at": [_P, _P, _I32, _P], "vq_db_write_avg": [_P, _P, _P],
    "vq_db_scores_grid": [_P, _P, _I32, _P, _I32, _P],
    "vq_db_loss_surface": [_P, _P, _I32, _P, _P, _I32, _P, _I32, _F64, _P],
    "vq_db_select": [_P, _F64, _F64, _pI64, _pI64, _pI64], "vq_db_select_fetch": [_P, _P, _I64, _P, _I64],
    "vq_db_select_rows": [_P, _F64, _F64, _P, _I64, _P, _I64, _pI64, _pI64, _pI64],
    "vq_db_round_layout": [_P, _P], "vq_db_query_round": [_P, _P, _I64, _I32, _F64, _F64], "vq_host_alloc": [_PP, _I64], "vq_host_free": [_P],
    "vq_db_topk": [_P, _I64, _P, _P, _pI64], "vq_db_min_score": [_P, _P, _I32, _pF64],
    "vq_resize_crop": [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P],
    "vq_resize_crop_planes": [_P, _I32, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P],
    "vq_tsn_create": [C.POINTER(TensorDesc), _I32, C.POINTER(LayerDesc), _I32, C.POINTER(ConvSegment), _I32, _P, _I64,
                      C.POINTER(InputDesc), _I32, _I32, _I32, _PP],
    "vq_tsn_destroy": [_P], "vq_tsn_set_stream": [_P, _P],
    "vq_tsn_forward": [_P, _P, _I32, _I32, _I32, _pF32, _P, _P],
    "vq_tsn_feat_devptr": [_P, _PP, _PP], "vq_tsn_read_tensor": [_P, _I32, _I32, _P],
    "vq_tsn_flops_per_crop": [_P, _pF64], "vq_tsn_launch_items": [_P, _P, _I32, _pI32], "vq_tsn_tuned_sizes": [_P, _P, _I32, _pI32],
    "vq_tsn_layer_tiles": [_P, _I32, _P, _I32], "vq_tsn_set_layer_tiles": [_P, _I32, _P, _I32],
    "vq_tsn_tile_tables": [_P, _P, _P, _I32, _pI32], "vq_tsn_get_tiles": [_P, _I32, _I32, _P, _I32], "vq_tsn_set_tiles": [_P, _I32, _I32, _P, _I32],
    "vq_tsn_tune": [_P, _I32, _I32], "vq_tsn_set_split": [_P, _I32, _I32], "vq_device_pool_trim": [],
    "vq_tsn_set_profile": [_P, _I32], "vq_tsn_set_profile_every": [_P, _I32], "vq_tsn_set_profile_split": [_P, _I32], "vq_tsn_layer_times": [_P, _P, _P, _I32],
    "vq_tvl1_default_params": [C.POINTER(Tvl1Params)],
    "vq_flow_create": [_I32, _I32, _I32, C.POINTER(Tvl1Params), _I32, _PP], "vq_flow_destroy": [_P],
    "vq_flow_levels": [_P, _pI32, _pI32, _I32],
    "vq_flow_tvl1": [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P],
    "vq_flow_last_timing": [_P, _pF64, _pI32],
    "vq_flow_warped": [_P, _P, _P, _I32, C.c_uint32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "vq_jpeg_info": [_P, _I64, _pI32, _pI32, _pI32],
    "vq_jpeg_create": [_I32, _I32, _I32, _I32, _PP], "vq_jpeg_destroy": [_P],
    "vq_jpeg_decode": [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P],
    "vq_jpeg_decode_files": [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P],
    "vq_jpeg_decode_path_list": [_P, C.c_char_p, C.c_int64, _I32, _I32, _I32, _I32, _P, _P, _P], "vq_jpeg_info_file": [C.c_char_p, _pI32, _pI32, _pI32],
    "vq_jpeg_crops": [_P, _I32, _I32, _P, _P],
    "vq_dev_malloc": [_PP, _I64, _I32], "vq_dev_free": [_P, _I32], "vq_stream_create": [_PP, _I32], "vq_stream_create_priority": [_PP, _I32, _I32], "vq_stream_destroy": [_P, _I32],
    "vq_stream_synchronize": [_P, _I32], "vq_dev_read": [_P, _P, _I64, _I32],
    "vq_flow_good_features": [_P, _P, _I32, _I32, _I32, C.c_float, C.c_float, _P, _P, _P],
    "vq_flow_ransac_homography": [_P, _P, _P, _P, _I32, _I32, C.c_float, _I32, C.c_uint32, _I32, _P, _P, _P, _P, _P],
    "vq_comm_unique_id": [_P], "vq_comm_init": [_I32, _I32, _P, _I32, _PP], "vq_comm_destroy": [_P],
    "vq_comm_info": [_P, _pI32, _pI32, _pI32],
    "vq_allgather_features": [_P, _P, _I64, _P, _P], "vq_allgather_scores": [_P, _P, _I64, _P, _P],
    "vq_broadcast_query": [_P, _P, _I64, _I32, _P],
}
_SPECIAL = {"vq_last_error": ([], C.c_char_p), "vq_abi_version": ([], C.c_int)}

_lib = None
_lock = threading.Lock()


def _preload_torch_hip():
    """If torch is importable, import it first so that ONE HIP runtime (torch's bundled
    libamdhip64.so.7) serves both torch and libvqamd; device pointers are then interchangeable.
    VQ_NO_TORCH=1 (set by the single-GPU command line before anything touches the library): the process will not use torch at all
    (tsn/devmem.py), the library runs on the system's HIP runtime and the 0.8 s import is saved."""
    global TORCHLESS
    if os.environ.get("VQ_NO_TORCH") == "1" and "torch" not in sys.modules:
        TORCHLESS = True
        return
    TORCHLESS = False
    try:
        import torch  # noqa: F401
    except Exception:
        pass


# Latched when the library is loaded (None before that): True = this process runs the library on the system's HIP runtime, WITHOUT
# torch; its device pointers and streams are then not interchangeable with those of a torch imported later (two runtimes).  Everything
# that decides between the library's own device plumbing and torch's reads THIS, never the environment again (tsn/devmem.native).
TORCHLESS = None


def require_torch_runtime(what: str):
    """For entry points that hand torch tensors to the library (sharded_db, features_tensor, bench): fail clearly in a process whose
    library was loaded torch-less instead of passing pointers between two HIP runtimes."""
    load()
    if TORCHLESS:
        raise RuntimeError("%s needs torch and the library on ONE HIP runtime, but libvqamd.so was loaded with VQ_NO_TORCH=1 (the "
                           "one-rank command line's torch-less mode) in this process; start a fresh process, or import torch before "
                           "the first use of the package" % what)


def load(path: str | None = None):
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or os.environ.get("VQ_AMD_LIB", LIB_PATH)
        if not os.path.exists(path):
            raise ImportError("libvqamd.so not found at %s -- build it with `python __graft_entry__.py build` "
                              "(there is no CPU fallback)" % path)
        _preload_torch_hip()
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (args, res) in _SPECIAL.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = ABI mismatch; fail loudly
            fn.argtypes, fn.restype = args, C.c_int
        if lib.vq_abi_version() != ABI_VERSION:
            raise ImportError("libvqamd.so ABI version %d != %d" % (lib.vq_abi_version(), ABI_VERSION))
        _lib = lib
        return lib


def check(rc: int):
    if rc != 0:
        raise VqError(rc, (load().vq_last_error() or b"").decode("utf-8", "replace"))


def call(name: str, *args):
    check(getattr(load(), name)(*args))


def exported_symbols():
    return sorted(list(SIGNATURES) + list(_SPECIAL))
