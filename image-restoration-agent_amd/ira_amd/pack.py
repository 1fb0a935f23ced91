"""Weight packer: reference state_dict -> flat device-ready blob for ``ira_model_create``.

PyTorch is used by callers only to READ a ``.pth`` (``torch.load(w)['params']``, demo.py:101-102);
everything here is numpy.  The blob is a table of contents + 256-byte-aligned arrays, laid out the way
the HIP kernels consume them (csrc/ira_kernels.hip):

* LayerNorm gain folded into the following 1x1 conv:  conv(LN(x)) = (W diag(g)) core(x) + W b_ln,
  where core(x) is x*rstd (BiasFree, restormer_arch.py:37-39) or (x-mu)*rstd (WithBias, :54-57).
* 1x1 weights fp16, hi and lo planes (``split16``), each in MFMA-fragment-major order
  ``[NB][K/16][64 lanes][8]`` (``frag_major``): one A-operand load of a wave is one contiguous 1 KB.
* GDFN hidden width padded to a multiple of 32 (127 -> 128, ...): x1 occupies channels [0, Hp), x2
  [Hp, 2Hp); padded rows/cols are zero so gelu(0)*0 = 0 flows through (restormer_arch.py:88-93).
* depthwise 3x3 weights fp32 tap-major ``[9][channels]``; dense 3x3 fp16 ``[2][9]`` x fragment-major.
* attention project_out kept fp32 ``[C][C]``: it is folded with the per-tile softmax on the device.
"""
from __future__ import annotations

import struct

import os

import numpy as np

from .config import RestormerConfig

MAGIC = b'IRAWGT1\x00'


def up32(n: int) -> int:
    return (n + 31) // 32 * 32


def _np(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t.astype(np.float32, copy=False)
    return t.detach().cpu().float().numpy()


def split16(a: np.ndarray) -> np.ndarray:
    """fp32/fp64 array -> stacked [hi, lo] fp16 planes with hi + lo ~= a to ~2^-22 relative.
    Every weight that feeds an MFMA is stored like this; the kernels issue one MFMA per plane."""
    a = np.asarray(a, dtype=np.float64)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float64)).astype(np.float16)
    return np.stack([hi, lo])


ROWPERM = np.array([16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3) for i in range(32)])


def frag_major(w: np.ndarray) -> np.ndarray:
    """[..., rows (multiple of 32), K (multiple of 16)] -> [..., NB, K/16, 64, 8]: MFMA A-fragment order of
    csrc/ira_kernels.hip.  Lane l = 32*h + r of fragment (nb, s) holds W[32nb + rowperm(r)][16s + 8h + j],
    j = 0..7, so that one A load of a wave is one contiguous 1 KB."""
    *lead, rows, k = w.shape
    nb, ks = rows // 32, k // 16
    a = w.reshape(*lead, nb, 32, ks, 2, 8)
    a = np.take(a, ROWPERM, axis=len(lead) + 1)                       # [.., nb, r, s, h, j]
    n = len(lead)
    a = np.transpose(a, list(range(n)) + [n, n + 2, n + 3, n + 1, n + 4])   # [.., nb, s, h, r, j]
    return np.ascontiguousarray(a.reshape(*lead, nb, ks, 64, 8))


def _pad_rows(a: np.ndarray, rows: int) -> np.ndarray:
    out = np.zeros((rows,) + a.shape[1:], dtype=a.dtype)
    out[:a.shape[0]] = a
    return out


def _hidden_remap(a: np.ndarray, hid: int, hp: int) -> np.ndarray:
    """[2*hid, ...] -> [2*hp, ...]: x1 rows to [0,hid), x2 rows to [hp, hp+hid); the rest zero."""
    out = np.zeros((2 * hp,) + a.shape[1:], dtype=a.dtype)
    out[:hid] = a[:hid]
    out[hp:hp + hid] = a[hid:]
    return out


def _conv3_pack(w: np.ndarray) -> np.ndarray:
    n, k = w.shape[:2]
    out = np.zeros((9, up32(n), k), dtype=np.float64)
    out[:, :n, :] = w.reshape(n, k, 9).transpose(2, 0, 1)
    return frag_major(split16(out))                                         # [2][9][NB][K/16][64][8]


def _dw_pack(w: np.ndarray) -> np.ndarray:
    ch = w.shape[0]
    return np.ascontiguousarray(w.reshape(ch, 9).T.astype(np.float32))     # [9][ch]


def _dw_unit_pack(w9: np.ndarray, hp: int) -> np.ndarray:
    """[9][2hp] tap-major depthwise weights -> [hp/16][2][18][8] for csrc/ira_ffn2.hip: everything ONE vector-work unit (16-channel
    chunk c of the GDFN hidden width, channel octet oc) needs is 144 contiguous floats: the nine taps of its 8 x1 channels
    (16c + 8oc ..), then the nine taps of its 8 gate channels (hp + 16c + 8oc ..).  The kernel walks them two "half taps" (16 floats,
    one s_load_dwordx16 at an immediate offset) per slot, x1 first: GELU needs x1 only, so its table look-ups run under the gate's
    FMAs."""
    a = w9.reshape(9, 2, hp // 16, 2, 8)                          # [tap][x1|x2][c][oc][e]
    return np.ascontiguousarray(a.transpose(2, 3, 1, 0, 4).reshape(hp // 16, 2, 18, 8).astype(np.float32))


def _dw_unit_bias(b: np.ndarray, hp: int) -> np.ndarray:
    a = b.reshape(2, hp // 16, 2, 8)                              # [x1|x2][c][oc][e]
    return np.ascontiguousarray(a.transpose(1, 2, 0, 3).reshape(hp // 16, 2, 16).astype(np.float32))


def pack_block(sd, p: str, dim: int, hid: int, cfg: RestormerConfig, out: dict, name: str):
    with_bias_ln = cfg.LayerNorm_type != 'BiasFree'
    hp = up32(hid)
    g1 = _np(sd[f'{p}.norm1.body.weight']).astype(np.float64)
    g2 = _np(sd[f'{p}.norm2.body.weight']).astype(np.float64)
    wq = _np(sd[f'{p}.attn.qkv.weight'])[:, :, 0, 0].astype(np.float64)          # [3C][C]
    wi = _np(sd[f'{p}.ffn.project_in.weight'])[:, :, 0, 0].astype(np.float64)    # [2hid][C]

    # IRA_ABLATE_LO_L12 (accuracy ablation, tools/ablate_lo_restormer.py): comma list of {qkv, pin, pout}: the lo planes of those 1x1 weights are
    # zeroed on the levels the fused kernels run (dim in {48, 96}) - numerically what a kernel that skips the lo-plane MFMAs computes
    # entries are a layer kind or kind@stage (stage = the state-dict prefix of the block's stage, e.g. pin@decoder_level1)
    abl = set(filter(None, os.environ.get('IRA_ABLATE_LO_L12', '').split(','))) if dim in (48, 96) else set()
    stage = p.split('.')[0]

    def planes(a, kind):
        sp = split16(a)
        if kind in abl or f'{kind}@{stage}' in abl:
            sp[1] = 0
        return frag_major(sp)

    out[f'{name}.qkv_w'] = planes(_pad_rows(wq * g1[None, :], up32(3 * dim)), 'qkv')
    out[f'{name}.pin_w'] = planes(_hidden_remap(wi * g2[None, :], hid, hp), 'pin')
    bq = np.zeros(3 * dim, np.float64)
    bi = np.zeros(2 * hid, np.float64)
    has_b = False
    if with_bias_ln:
        bq += wq @ _np(sd[f'{p}.norm1.body.bias']).astype(np.float64)
        bi += wi @ _np(sd[f'{p}.norm2.body.bias']).astype(np.float64)
        has_b = True
    if cfg.bias:
        bq += _np(sd[f'{p}.attn.qkv.bias'])
        bi += _np(sd[f'{p}.ffn.project_in.bias'])
        has_b = True
    if has_b:
        out[f'{name}.qkv_b'] = _pad_rows(bq.astype(np.float32), up32(3 * dim))
        out[f'{name}.pin_b'] = _hidden_remap(bi.astype(np.float32), hid, hp)

    out[f'{name}.dw1_w'] = _dw_pack(_np(sd[f'{p}.attn.qkv_dwconv.weight']))
    out[f'{name}.dw2_w'] = _dw_pack(_hidden_remap(_np(sd[f'{p}.ffn.dwconv.weight']), hid, hp))
    out[f'{name}.dw2_u'] = _dw_unit_pack(out[f'{name}.dw2_w'], hp)
    out[f'{name}.temp'] = _np(sd[f'{p}.attn.temperature']).reshape(-1).astype(np.float32)
    out[f'{name}.wo'] = np.ascontiguousarray(_np(sd[f'{p}.attn.project_out.weight'])[:, :, 0, 0])
    wout = np.zeros((up32(dim), hp), np.float64)
    wout[:dim, :hid] = _np(sd[f'{p}.ffn.project_out.weight'])[:, :, 0, 0]
    out[f'{name}.pout_w'] = planes(wout, 'pout')
    if cfg.bias:
        out[f'{name}.dw1_b'] = _np(sd[f'{p}.attn.qkv_dwconv.bias']).astype(np.float32)
        out[f'{name}.dw2_b'] = _hidden_remap(_np(sd[f'{p}.ffn.dwconv.bias']).astype(np.float32), hid, hp)
        out[f'{name}.dw2_ub'] = _dw_unit_bias(out[f'{name}.dw2_b'], hp)
        out[f'{name}.wo_b'] = _pad_rows(_np(sd[f'{p}.attn.project_out.bias']).astype(np.float32), up32(dim))
        out[f'{name}.pout_b'] = _pad_rows(_np(sd[f'{p}.ffn.project_out.bias']).astype(np.float32), up32(dim))


def pack_arrays(sd, cfg: RestormerConfig) -> dict:
    """name -> numpy array, in the device layouts (see module docstring)."""
    d = cfg.dim
    out: dict = {}
    out['embed_w'] = np.ascontiguousarray(_np(sd['patch_embed.proj.weight']))

    def stage(name, n, dim):
        for i in range(n):
            pack_block(sd, f'{name}.{i}', dim, cfg.hidden(dim), cfg, out, f'{name}.{i}')

    nb = cfg.num_blocks
    stage('encoder_level1', nb[0], d)
    stage('encoder_level2', nb[1], d * 2)
    stage('encoder_level3', nb[2], d * 4)
    stage('latent', nb[3], d * 8)
    stage('decoder_level3', nb[2], d * 4)
    stage('decoder_level2', nb[1], d * 2)
    stage('decoder_level1', nb[0], d * 2)
    stage('refinement', cfg.num_refinement_blocks, d * 2)
    for src, dst in [('down1_2.body.0.weight', 'down1_2_w'), ('down2_3.body.0.weight', 'down2_3_w'),
                     ('down3_4.body.0.weight', 'down3_4_w'), ('up4_3.body.0.weight', 'up4_3_w'),
                     ('up3_2.body.0.weight', 'up3_2_w'), ('up2_1.body.0.weight', 'up2_1_w'),
                     ('output.weight', 'output_w')]:
        out[dst] = _conv3_pack(_np(sd[src]))
    for src, dst in [('reduce_chan_level3', 'reduce3'), ('reduce_chan_level2', 'reduce2')]:
        w = _np(sd[f'{src}.weight'])[:, :, 0, 0]
        out[f'{dst}_w'] = frag_major(split16(_pad_rows(w, up32(w.shape[0]))))
        if cfg.bias:
            out[f'{dst}_b'] = _pad_rows(_np(sd[f'{src}.bias']).astype(np.float32), up32(w.shape[0]))
    if cfg.bias:
        out['output_b'] = _pad_rows(_np(sd['output.bias']).astype(np.float32), up32(cfg.out_channels))
    return out


def pack_blob(sd, cfg: RestormerConfig) -> bytes:
    return blob_from_arrays(pack_arrays(sd, cfg))


def blob_from_arrays(arrays: dict) -> bytes:
    """Container: magic, table of contents (128-byte entries: name, offset, nbytes), 256-byte-aligned arrays."""
    names = list(arrays.keys())
    header_bytes = 16 + 128 * len(names)
    off = (header_bytes + 255) // 256 * 256
    toc = []
    chunks = []
    for n in names:
        a = np.ascontiguousarray(arrays[n])
        nbytes = a.nbytes
        toc.append((n, off, nbytes))
        chunks.append((off, a.tobytes()))
        off = (off + nbytes + 255) // 256 * 256
    buf = bytearray(off)
    buf[0:8] = MAGIC
    struct.pack_into('<II', buf, 8, len(names), 0)
    pos = 16
    for n, o, nb in toc:
        nb_name = n.encode()
        assert len(nb_name) < 112, n
        buf[pos:pos + len(nb_name)] = nb_name
        struct.pack_into('<QQ', buf, pos + 112, o, nb)
        pos += 128
    for o, b in chunks:
        buf[o:o + len(b)] = b
    return bytes(buf)
