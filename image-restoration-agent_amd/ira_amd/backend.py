"""Python handle on one libira_hip model (one GPU, one stream).

Replaces, for the Restormer tool, what the reference builds per process in demo.py:92-103 /
model_serving.py:101-124 (instantiate arch, load ``checkpoint['params']``, ``.to(device)``, ``.eval()``)
and what it calls per image (``model(input_)`` / the tiler loop, demo.py:129-153).
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import _capi
from ._capi import IraConfig, IraError, check
from .config import RestormerConfig
from .pack import pack_blob


def _cfg_struct(cfg: RestormerConfig) -> IraConfig:
    s = IraConfig()
    s.inp_channels, s.out_channels, s.dim = cfg.inp_channels, cfg.out_channels, cfg.dim
    for i in range(4):
        s.num_blocks[i] = cfg.num_blocks[i]
        s.heads[i] = cfg.heads[i]
        s.hidden[i] = cfg.hidden(cfg.dim * 2 ** i)     # int(dim*ffn_expansion_factor), restormer_arch.py:80
    s.num_refinement_blocks = cfg.num_refinement_blocks
    s.ln_with_bias = 0 if cfg.LayerNorm_type == 'BiasFree' else 1
    s.conv_bias = 1 if cfg.bias else 0
    s.dual_pixel_task = 1 if cfg.dual_pixel_task else 0
    return s


class _Backend:
    """Shared handle plumbing: one libira_hip model on one MI355X (one HIP stream)."""

    def _create(self, device: int, create_fn, cfg_struct, blob: bytes, what: str):
        self._lib = _capi.load_library()
        self.device = int(device)
        self._h = C.c_void_p(None)
        if _capi.device_count() <= 0:
            raise IraError("no gfx950 (MI355X) device visible; this backend has no CPU fallback")
        h = C.c_void_p(None)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rc = getattr(self._lib, create_fn)(self.device, C.byref(cfg_struct), buf, len(blob), C.byref(h))
        check(rc, None, what)
        self._h = h

    # -- life cycle
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.ira_model_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def _ck(self, rc, what):
        check(rc, self._h, what)

    # -- network
    def forward(self, x_nchw: np.ndarray) -> np.ndarray:
        """``model(input_)`` (demo.py:131): float32 (B,C,H,W) in, float32 (B,C,H,W) out."""
        x = np.ascontiguousarray(x_nchw, dtype=np.float32)
        if x.ndim != 4:
            raise ValueError("expected (B,C,H,W)")
        B, Cc, H, W = x.shape
        s = int(getattr(self, 'scale', 1))
        out = np.empty((B, Cc, H * s, W * s), np.float32)
        self._ck(self._lib.ira_forward(self._h, x.ctypes.data, B, Cc, H, W, out.ctypes.data), 'ira_forward')
        return out

    def forward_device(self, in_ptr: int, B: int, H: int, W: int, out_ptr: int):
        """float32 NHWC device tensors (raw pointers, e.g. torch.Tensor.data_ptr())."""
        self._ck(self._lib.ira_forward_device(self._h, in_ptr, B, H, W, out_ptr), 'ira_forward_device')

    # -- strip-parallel whole frame (host side: ira_amd/dist.py::StripParallelRestorer)
    def strip_buffer_bytes(self, Wp: int):
        e, r = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._lib.ira_strip_buffer_bytes(self._h, int(Wp), C.byref(e), C.byref(r)), 'ira_strip_buffer_bytes')
        return e.value, r.value

    def strip_configure(self, comm):
        """comm: _capi.IraStripComm (kept alive by the caller) or None."""
        self._ck(self._lib.ira_strip_configure(self._h, C.byref(comm) if comm is not None else None), 'ira_strip_configure')

    def strip_forward_device(self, in_ptr: int, H_local: int, W: int, halo_top: int, halo_bot: int, out_ptr: int):
        self._ck(self._lib.ira_strip_forward_device(self._h, in_ptr, H_local, W, halo_top, halo_bot, out_ptr), 'ira_strip_forward_device')

    # -- harness
    def restore(self, img_u8: np.ndarray, tile=None, tile_overlap: int = 32) -> np.ndarray:
        """uint8 (h,w,c) -> uint8 (h,w,c): pad -> whole/tiled forward -> average -> clamp -> crop -> uint8
        (demo.py:120-161).  ``tile=None`` is the whole-image mode (demo.py:129-131)."""
        img = np.ascontiguousarray(img_u8)
        if img.dtype != np.uint8 or img.ndim != 3:
            raise ValueError("expected a uint8 (h,w,c) array")
        h, w, c = img.shape
        s = int(getattr(self, 'scale', 1))
        out = np.empty((h * s, w * s, c), np.uint8)
        rc = self._lib.ira_restore_u8(self._h, img.ctypes.data, h, w, c, int(tile) if tile else 0,
                                      int(tile_overlap), out.ctypes.data)
        self._ck(rc, 'ira_restore_u8')
        return out

    def plan(self, h: int, w: int, tile, overlap: int):
        """(Hp, Wp, tile_h, tile_w, n_rows, n_cols, scale) of this model's harness (SwinIR: always-pad rule and upscale)."""
        v = [C.c_int(0) for _ in range(7)]
        rc = self._lib.ira_model_tile_plan(self._h, h, w, int(tile) if tile else 0, int(overlap), *[C.byref(x) for x in v])
        self._ck(rc, 'ira_model_tile_plan')
        return tuple(x.value for x in v)

    def restore_padcrop(self, img_u8: np.ndarray, tile_size=None, tile_pad: int = 32, padding_size: int = 64) -> np.ndarray:
        """The reference's other tiler (xrestormer_model.py:53-136: reflect-pad to ``padding_size``, non-overlapping
        ``tile_size`` cores with a ``tile_pad`` halo, crop): uint8 (h,w,c) -> uint8 (h,w,c)."""
        img = np.ascontiguousarray(img_u8)
        if img.dtype != np.uint8 or img.ndim != 3:
            raise ValueError("expected a uint8 (h,w,c) array")
        h, w, c = img.shape
        s = int(getattr(self, 'scale', 1))
        out = np.empty((h * s, w * s, c), np.uint8)
        rc = self._lib.ira_restore_padcrop_u8(self._h, img.ctypes.data, h, w, c, int(tile_size) if tile_size else 0,
                                              int(tile_pad), int(padding_size), out.ctypes.data)
        self._ck(rc, 'ira_restore_padcrop_u8')
        return out

    def restore_tiles_device(self, frame_ptr: int, h: int, w: int, c: int, tile, overlap: int,
                             first_tile: int, n_tiles: int, out_ptr: int):
        rc = self._lib.ira_restore_tiles_device(self._h, frame_ptr, h, w, c, int(tile) if tile else 0, int(overlap),
                                                first_tile, n_tiles, out_ptr)
        self._ck(rc, 'ira_restore_tiles_device')

    def blend_tiles_device(self, tiles_ptr: int, h: int, w: int, c: int, tile, overlap: int, out_ptr: int):
        rc = self._lib.ira_blend_tiles_device(self._h, tiles_ptr, h, w, c, int(tile) if tile else 0, int(overlap), out_ptr)
        self._ck(rc, 'ira_blend_tiles_device')

    # tensor-taking forms used by ira_amd.dist (torch is plumbing: device memory only)
    def restore_tiles_t(self, frame, h, w, c, tile, overlap, first_tile, n_tiles, out):
        assert frame.is_cuda and out.is_cuda and frame.is_contiguous() and out.is_contiguous()
        self.restore_tiles_device(frame.data_ptr(), h, w, c, tile, overlap, first_tile, n_tiles, out.data_ptr())

    def blend_tiles_t(self, tiles, h, w, c, tile, overlap, out):
        assert tiles.is_cuda and out.is_cuda and tiles.is_contiguous() and out.is_contiguous()
        self.blend_tiles_device(tiles.data_ptr(), h, w, c, tile, overlap, out.data_ptr())

    # -- measurement
    def set_max_batch(self, n: int):
        self._ck(self._lib.ira_set_max_batch(self._h, int(n)), 'ira_set_max_batch')

    def set_fused_ffn(self, on: bool):
        self._ck(self._lib.ira_set_fused_ffn(self._h, 1 if on else 0), 'ira_set_fused_ffn')

    # -- RCCL communicator of this handle (csrc/ira_comm.hip)
    has_comm = False

    def comm_init(self, uid: bytes, rank: int, world: int):
        if len(uid) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        buf = (C.c_char * 128).from_buffer_copy(uid)
        self._ck(self._lib.ira_comm_init(self._h, buf, int(rank), int(world)), 'ira_comm_init')
        self.has_comm = True

    def comm_destroy(self):
        if self.has_comm:
            self._ck(self._lib.ira_comm_destroy(self._h), 'ira_comm_destroy')
            self.has_comm = False

    def restore_frame_distributed_t(self, frame, tile, overlap, out):
        assert frame.is_cuda and out.is_cuda and frame.is_contiguous() and out.is_contiguous()
        h, w, c = frame.shape
        rc = self._lib.ira_restore_frame_distributed(self._h, frame.data_ptr(), h, w, c, int(tile) if tile else 0, int(overlap), out.data_ptr())
        self._ck(rc, 'ira_restore_frame_distributed')

    def strip_use_comm(self, Wp: int):
        self._ck(self._lib.ira_strip_use_comm(self._h, int(Wp)), 'ira_strip_use_comm')

    def strip_restore_u8_t(self, frame, out):
        assert frame.is_cuda and out.is_cuda and frame.is_contiguous() and out.is_contiguous()
        h, w, c = frame.shape
        self._ck(self._lib.ira_strip_restore_u8_device(self._h, frame.data_ptr(), h, w, c, out.data_ptr()), 'ira_strip_restore_u8_device')

    def set_weight_planes(self, planes: int):
        """ira_set_weight_planes: 1 = hi plane alone in the 1x1 layers that have a kernel for it (the state of a Restormer / X-Restormer /
        SwinIR handle after create, for C and Python callers alike), 2 = hi + lo fp16 weight planes everywhere.  Either way round on the
        same handle: the lo planes are kept in a device-side copy."""
        self._ck(self._lib.ira_set_weight_planes(self._h, int(planes)), 'ira_set_weight_planes')

    @property
    def weight_planes(self) -> int:
        v = C.c_int(0)
        self._ck(self._lib.ira_get_weight_planes(self._h, C.byref(v)), 'ira_get_weight_planes')
        return v.value

    def set_ffn_generation(self, gen: int):
        self._ck(self._lib.ira_set_ffn_generation(self._h, int(gen)), 'ira_set_ffn_generation')

    def sync(self):
        self._ck(self._lib.ira_sync(self._h), 'ira_sync')

    def timer_start(self):
        self._ck(self._lib.ira_timer_start(self._h), 'ira_timer_start')

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._ck(self._lib.ira_timer_stop(self._h, C.byref(ms)), 'ira_timer_stop')
        return float(ms.value)

    def profile_enable(self, on: bool):
        self._ck(self._lib.ira_profile_enable(self._h, 1 if on else 0), 'ira_profile_enable')

    def profile(self) -> dict:
        s = self._lib.ira_profile_json(self._h)
        return json.loads(s.decode()) if s else {}

    def memory_bytes(self) -> int:
        n = C.c_size_t(0)
        self._ck(self._lib.ira_memory_bytes(self._h, C.byref(n)), 'ira_memory_bytes')
        return int(n.value)


class RestormerBackend(_Backend):
    """One Restormer on one MI355X.  ``state_dict``: the reference's name -> tensor mapping
    (torch tensors or numpy arrays), e.g. ``torch.load(path)['params']``."""

    def __init__(self, cfg: RestormerConfig, state_dict, device: int = 0, weight_planes: int = 1):
        """``weight_planes``: 1 (default) = the 1x1 layers of the levels with 192 / 384 channels (levels 3-4 of the agent's presets) run on
        plain fp16 weights - what ira_model_create itself leaves the handle in -, 2 = hi + lo planes everywhere (ira_set_weight_planes)."""
        self.cfg = cfg
        self._create(device, 'ira_model_create', _cfg_struct(cfg), pack_blob(state_dict, cfg), 'ira_model_create')
        self.set_weight_planes(weight_planes)


class NAFNetBackend(_Backend):
    """One NAFNet on one MI355X (NAFNet_arch.py:249-326).  ``forward`` takes any H, W: the network
    pads to a multiple of 2**len(enc_blk_nums) with zeros and crops (check_image_size, :356-361)."""

    def __init__(self, cfg, state_dict, device: int = 0):
        from .nafnet import pack_nafnet_blob, nafnet_cfg_struct
        self.cfg = cfg
        self._create(device, 'ira_nafnet_create', nafnet_cfg_struct(cfg), pack_nafnet_blob(state_dict, cfg), 'ira_nafnet_create')


class DnCNNBackend(_Backend):
    """The build-defined DnCNN (BASELINE config 1; no reference code exists: see ira_amd/dncnn.py)."""

    def __init__(self, cfg, state_dict, device: int = 0):
        from .dncnn import pack_dncnn_blob, dncnn_cfg_struct
        self.cfg = cfg
        self._create(device, 'ira_dncnn_create', dncnn_cfg_struct(cfg), pack_dncnn_blob(state_dict, cfg), 'ira_dncnn_create')


class MPRNetBackend(_Backend):
    """One MPRNet on one MI355X (AgenticIR/executor/denoising/tools/MPRNet).  ``forward``: float32 (B,3,H,W), H and W
    multiples of 8, -> the stage-3 image (restored[0] of the reference's list).  ``restore(img, tile=None)``: the demo.py:64-87 harness."""

    def __init__(self, cfg, state_dict, device: int = 0):
        from .mprnet import pack_mprnet_blob, mprnet_cfg_struct, prelu_value
        self.cfg = cfg
        self._create(device, 'ira_mprnet_create', mprnet_cfg_struct(cfg, prelu_value(state_dict)), pack_mprnet_blob(state_dict, cfg),
                     'ira_mprnet_create')


class RRDBNetBackend(_Backend):
    """One RRDBNet (Real-ESRGAN generator) on one MI355X (DiffBIR/utils/realesrgan/rrdbnet.py).  ``forward``: float32
    (B,C,H,W) -> (B,C,H*scale,W*scale).  ``restore_padcrop(img, tile_size, tile_pad, padding_size=mod_scale)``: the tensor path of
    RealESRGANer (realesrganer.py:71-183) on uint8 images."""

    def __init__(self, cfg, state_dict, device: int = 0):
        from .rrdbnet import pack_rrdbnet_blob, rrdbnet_cfg_struct
        self.cfg = cfg
        self.scale = cfg.scale
        self._create(device, 'ira_rrdbnet_create', rrdbnet_cfg_struct(cfg), pack_rrdbnet_blob(state_dict, cfg), 'ira_rrdbnet_create')


class SwinIRBackend(_Backend):
    """One SwinIR on one MI355X (SwinIR/models/network_swinir.py).  ``forward``: float32 (B,C,H,W) ->
    (B,C,H*upscale,W*upscale), any H, W.  ``restore``: the reference harness (SwinIR/inference.py:69-90,294-322: always-pad by
    flip-concat, whole image or overlap-average tiler, clamp, round): uint8 (h,w,c) -> uint8 (h*s, w*s, c)."""

    def __init__(self, cfg, state_dict, device: int = 0, weight_planes: int = 1):
        """``weight_planes``: 1 (default) = the transformer layers' 1x1s run on plain fp16 weights (the lo planes are zeroed in the handle;
        frame-sized launches of K = 192 / 256 layers skip them), 2 = hi + lo planes everywhere.  Measured on the reference-generated goldens
        and at the benchmark geometry against the oracle: |dPSNR| <= 6e-5 dB.  1 is also the state ira_swinir_create leaves."""
        from .swinir import pack_swinir_blob, swinir_cfg_struct
        self.cfg = cfg
        self.scale = cfg.upscale
        self._create(device, 'ira_swinir_create', swinir_cfg_struct(cfg), pack_swinir_blob(state_dict, cfg), 'ira_swinir_create')
        self.set_weight_planes(weight_planes)


class XRestormerBackend(_Backend):
    """One X-Restormer on one MI355X (xrestormer_arch.py:324-422).  ``forward`` needs H, W multiples of 64;
    ``restore_padcrop(img, tile_size, tile_pad, padding_size=64)`` is its harness (xrestormer_model.py:53-136)."""

    def __init__(self, cfg, state_dict, device: int = 0, weight_planes: int = 1):
        from .xrestormer import pack_xrestormer_blob, xrestormer_cfg_struct
        self.cfg = cfg
        self.scale = cfg.scale
        self._create(device, 'ira_xrestormer_create', xrestormer_cfg_struct(cfg), pack_xrestormer_blob(state_dict, cfg), 'ira_xrestormer_create')
        self.set_weight_planes(weight_planes)       # 1: levels 3-4 (C = 192 / 384) on plain fp16 weights, as RestormerBackend
