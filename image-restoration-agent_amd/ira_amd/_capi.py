"""ctypes binding of libira_hip.so (include/ira.h).  Thin: argument marshalling and error raising only.

There is no CPU fallback: if the shared library is missing, or no gfx950 device is visible, the
product path raises (``IraError``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libira_hip.so')


class IraError(RuntimeError):
    pass


class IraConfig(C.Structure):
    _fields_ = [
        ('inp_channels', C.c_int32), ('out_channels', C.c_int32), ('dim', C.c_int32),
        ('num_blocks', C.c_int32 * 4), ('num_refinement_blocks', C.c_int32),
        ('heads', C.c_int32 * 4), ('hidden', C.c_int32 * 4),
        ('ln_with_bias', C.c_int32), ('conv_bias', C.c_int32), ('dual_pixel_task', C.c_int32),
    ]


class IraXrestormerConfig(C.Structure):
    _fields_ = [('base', IraConfig), ('spatial_heads', C.c_int32 * 4), ('window_size', C.c_int32), ('overlap_win', C.c_int32 * 4),
                ('spatial_dim_head', C.c_int32), ('scale', C.c_int32)]


NAFNET_MAX_LEVELS = 6


class IraNafnetConfig(C.Structure):
    _fields_ = [
        ('img_channel', C.c_int32), ('width', C.c_int32), ('n_levels', C.c_int32),
        ('enc_blk_nums', C.c_int32 * NAFNET_MAX_LEVELS), ('middle_blk_num', C.c_int32),
        ('dec_blk_nums', C.c_int32 * NAFNET_MAX_LEVELS),
    ]


SWINIR_MAX_LAYERS = 12


class IraSwinirConfig(C.Structure):
    _fields_ = [
        ('upscale', C.c_int32), ('in_chans', C.c_int32), ('window_size', C.c_int32), ('img_range', C.c_float),
        ('n_layers', C.c_int32), ('depths', C.c_int32 * SWINIR_MAX_LAYERS), ('embed_dim', C.c_int32),
        ('num_heads', C.c_int32 * SWINIR_MAX_LAYERS), ('mlp_hidden', C.c_int32), ('upsampler', C.c_int32), ('resi_3conv', C.c_int32),
    ]


class IraDncnnConfig(C.Structure):
    _fields_ = [('in_nc', C.c_int32), ('nc', C.c_int32), ('nb', C.c_int32)]


class IraMprnetConfig(C.Structure):
    _fields_ = [('in_c', C.c_int32), ('out_c', C.c_int32), ('n_feat', C.c_int32), ('scale_unetfeats', C.c_int32),
                ('scale_orsnetfeats', C.c_int32), ('num_cab', C.c_int32), ('reduction', C.c_int32), ('prelu', C.c_float)]


class IraRrdbnetConfig(C.Structure):
    _fields_ = [('num_in_ch', C.c_int32), ('num_out_ch', C.c_int32), ('scale', C.c_int32), ('num_feat', C.c_int32), ('num_block', C.c_int32),
                ('num_grow_ch', C.c_int32)]


STRIP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)


class IraStripComm(C.Structure):
    _fields_ = [('ctx', C.c_void_p), ('exchange', STRIP_FN), ('allgather', STRIP_FN), ('rank', C.c_int32), ('world', C.c_int32),
                ('send_top', C.c_void_p), ('recv_top', C.c_void_p), ('send_bot', C.c_void_p), ('recv_bot', C.c_void_p),
                ('edge_capacity', C.c_uint64), ('gather_send', C.c_void_p), ('gather_recv', C.c_void_p), ('gather_capacity', C.c_uint64)]


_vp, _i, _l, _sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
_pi = C.POINTER(C.c_int)

# name -> (restype, argtypes): every symbol include/ira.h declares
SIGNATURES = {
    'ira_device_count': (_i, [_pi]),
    'ira_model_create': (_i, [_i, C.POINTER(IraConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_xrestormer_create': (_i, [_i, C.POINTER(IraXrestormerConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_nafnet_create': (_i, [_i, C.POINTER(IraNafnetConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_dncnn_create': (_i, [_i, C.POINTER(IraDncnnConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_swinir_create': (_i, [_i, C.POINTER(IraSwinirConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_mprnet_create': (_i, [_i, C.POINTER(IraMprnetConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_rrdbnet_create': (_i, [_i, C.POINTER(IraRrdbnetConfig), _vp, _sz, C.POINTER(_vp)]),
    'ira_model_destroy': (_i, [_vp]),
    'ira_last_error': (C.c_char_p, [_vp]),
    'ira_sync': (_i, [_vp]),
    'ira_forward': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'ira_forward_device': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'ira_strip_configure': (_i, [_vp, C.POINTER(IraStripComm)]),
    'ira_strip_buffer_bytes': (_i, [_vp, _i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'ira_strip_forward_device': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'ira_comm_unique_id': (_i, [_vp]),
    'ira_comm_init': (_i, [_vp, _vp, _i, _i]),
    'ira_comm_destroy': (_i, [_vp]),
    'ira_comm_rank': (_i, [_vp, _pi, _pi]),
    'ira_restore_frame_distributed': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ira_strip_use_comm': (_i, [_vp, _i]),
    'ira_strip_restore_u8_device': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'ira_tile_plan': (_i, [_i, _i, _i, _i, _pi, _pi, _pi, _pi, _pi, _pi]),
    'ira_model_tile_plan': (_i, [_vp, _i, _i, _i, _i, _pi, _pi, _pi, _pi, _pi, _pi, _pi]),
    'ira_restore_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ira_restore_padcrop_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ira_restore_tiles_device': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ira_blend_tiles_device': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ira_brighten_u8': (_i, [_i, _i, _vp, _i, _i, _vp]),
    'ira_brighten_u8_device': (_i, [_i, _vp, _i, _vp, _i, _i, _vp]),
    'ira_set_max_batch': (_i, [_vp, _i]),
    'ira_timer_start': (_i, [_vp]),
    'ira_timer_stop': (_i, [_vp, C.POINTER(C.c_float)]),
    'ira_profile_enable': (_i, [_vp, _i]),
    'ira_profile_json': (C.c_char_p, [_vp]),
    'ira_memory_bytes': (_i, [_vp, C.POINTER(_sz)]),
    'ira_op_pw_kres': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    'ira_op_pw_nres': (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_dwconv_qkv': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_dwconv_gate': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_gram': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    'ira_op_attn_mix': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    'ira_op_conv3x3': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    'ira_op_ffn_fused': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_ffn2': (_i, [_vp, _vp, _vp, _vp, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_ffn2_planes': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_set_ffn_generation': (_i, [_vp, _i]),
    'ira_set_weight_planes': (_i, [_vp, _i]),
    'ira_get_weight_planes': (_i, [_vp, _vp]),
    'ira_op_qkv_fused': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    'ira_op_qkv_fused_gen': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    'ira_set_fused_ffn': (_i, [_vp, _i]),
    'ira_op_embed': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    'ira_op_concat2': (_i, [_vp, _vp, _i, _vp, _i, _vp, _l]),
    'ira_op_pw_kres_eps': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float]),
    'ira_op_naf_pw': (_i, [_vp, _i, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ira_op_naf_dw_sg': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _pi]),
    'ira_op_ocab_attn': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    'ira_op_swin_attn': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    'ira_op_ln_affine': (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, C.c_float]),
    'ira_op_pw_kres_ex': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _i, _i]),
    'ira_png_info': (_i, [_vp, C.c_size_t, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'ira_png_decode': (_i, [_vp, C.c_size_t, _vp, C.c_size_t]),
    'ira_op_swin_proj_mlp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, C.c_float, _i]),
    'ira_op_swin_mlp': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, C.c_float, _i]),
    'ira_op_naf_tail': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    'ira_op_naf_sca': (_i, [_vp, _vp, _i, _l, _vp, _vp, _vp, _i, _i]),
}

_lib = None


def load_library(path: str | None = None):
    """dlopen libira_hip.so and attach prototypes. Raises IraError when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get('IRA_HIP_LIB', LIB_PATH)
    if not os.path.exists(p):
        raise IraError(f"{p} not found: build it with `make -C image-restoration-agent_amd/csrc` "
                       "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:   # missing libamdhip64 etc.
        raise IraError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def last_error(handle=None) -> str:
    lib = load_library()
    s = lib.ira_last_error(handle)
    return s.decode(errors='replace') if s else ''


def check(rc: int, handle=None, what: str = ''):
    if rc != 0:
        raise IraError(f"{what or 'libira_hip'} failed (code {rc}): {last_error(handle)}")


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.ira_device_count(C.byref(n))
    if rc != 0:
        return 0
    return n.value


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0); 128 bytes to carry to the other ranks."""
    lib = load_library()
    buf = (C.c_char * 128)()
    check(lib.ira_comm_unique_id(buf), None, 'ira_comm_unique_id')
    return bytes(buf)


def tile_plan(h: int, w: int, tile, overlap: int):
    """(Hp, Wp, tile_h, tile_w, n_rows, n_cols) of the reference tiler (demo.py:122-141)."""
    lib = load_library()
    v = [C.c_int(0) for _ in range(6)]
    rc = lib.ira_tile_plan(h, w, int(tile) if tile else 0, int(overlap), *[C.byref(x) for x in v])
    check(rc, None, 'ira_tile_plan')
    return tuple(x.value for x in v)


BRIGHTEN_TOOLS = {'constant_shift': 0, 'gamma_correction': 1, 'histogram_equalization': 2}     # IRA_BRIGHTEN_* of include/ira.h


def brighten_u8(img, tool_name: str, device: int = 0):
    """ira_brighten_u8 on a uint8 (h,w,3) RGB numpy image: what BrighteningTool._invoke does between imread and imwrite
    (AgenticIR/executor/brightening/__init__.py:21-29).  Raises IraError without a gfx950 device: there is no CPU path."""
    import numpy as np
    lib = load_library()
    if tool_name not in BRIGHTEN_TOOLS:
        raise KeyError(f"brightening tools are {sorted(BRIGHTEN_TOOLS)}, not {tool_name!r}")
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("brighten_u8 expects a uint8 (h,w,3) RGB image")
    out = np.empty_like(a)
    rc = lib.ira_brighten_u8(int(device), BRIGHTEN_TOOLS[tool_name], a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data)
    check(rc, None, 'ira_brighten_u8')
    return out


def decode_png(data: bytes):
    """ira_png_info + ira_png_decode: the raster as stored - uint8 (h,w,channels), channels 1 grey, 2 grey + alpha, 3 RGB (palette
    images expanded), 4 RGBA.  Raises IraError (with the library's reason) for what the native decoder leaves to PIL: 16-bit samples,
    Adam7, 1/2/4-bit rasters, and anything that is not a PNG stream."""
    import numpy as np
    lib = load_library()
    buf = bytes(data)
    h, w, c = C.c_int(0), C.c_int(0), C.c_int(0)
    check(lib.ira_png_info(buf, len(buf), C.byref(h), C.byref(w), C.byref(c)), None, 'ira_png_info')
    out = np.empty((h.value, w.value, c.value), np.uint8)
    check(lib.ira_png_decode(buf, len(buf), out.ctypes.data, out.nbytes), None, 'ira_png_decode')
    return out

