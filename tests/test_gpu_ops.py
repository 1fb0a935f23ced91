"""Per-kernel parity on a real MI355X: every hand-written HIP kernel, called through the C ABI
(include/ira.h, ira_op_*), against a plain torch CPU fp32 statement of the same op on the same
(fp16-rounded) inputs.  Reference ops: restormer_arch.py (cited per test).

Tolerances: outputs stored as fp16 are compared with rtol 2e-3 (fp16 has 2^-11 relative spacing);
fp32 outputs of the weight contractions with 2e-5: weights travel as fp16 hi + lo planes, activations
are fp16-rounded identically on both sides, so only fp32 summation order differs.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def r16(t):
    return t.half().float()


@pytest.fixture(scope='module')
def ctx():
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.weights import seeded_state_dict
    cfg = reduced_config('BiasFree')
    be = RestormerBackend(cfg, seeded_state_dict(cfg, 1), device=0, weight_planes=2)      # the op tests check hi + lo kernels; hi-plane kernels: _hi_plane_ctx
    yield be
    be.close()


def dev(t):
    return t.contiguous().cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def rowperm_pad(w, rows):
    out = torch.zeros(rows, w.shape[1], dtype=w.dtype)
    out[:w.shape[0]] = w
    return out


def up32(n):
    return (n + 31) // 32 * 32


def split16(w):
    """fp32 [..., rows, K] -> device layout of ira_amd.pack: stacked [hi, lo] fp16 planes (hi + lo ~= w to
    ~2^-22), each in MFMA-fragment-major order [..., NB, K/16, 64, 8]."""
    from ira_amd.pack import frag_major, split16 as _s16
    return torch.from_numpy(frag_major(_s16(w.double().numpy())))


def close(a, b, rtol, atol_scale=1e-3):
    a = a.double(); b = b.double()
    assert torch.isfinite(a).all(), f"{int((~torch.isfinite(a)).sum())}/{a.numel()} non-finite values in the result"
    atol = atol_scale * float(b.abs().max()) + 1e-6
    err = (a - b).abs()
    bad = err > (atol + rtol * b.abs())
    assert not bad.any(), f"max err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e}), {int(bad.sum())}/{bad.numel()} bad"


def ln_core(x, mode):
    if mode == 0:
        return x
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, keepdim=True, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    return x * rstd if mode == 1 else (x - mu) * rstd


@pytest.mark.parametrize('K,N,ln,in_f32,T,P,with_bias', [
    (48, 144, 1, True, 2, 77, False),      # LN(BiasFree)+qkv at level 1 (restormer_arch.py:37-39,105)
    (96, 512, 2, True, 1, 64, True),       # LN(WithBias)+project_in, padded hidden 255->256 (:54-57,82)
    (16, 48, 1, True, 3, 33, False),       # reduced config
    (384, 1152, 1, True, 1, 40, False),    # latent level
    (192, 576, 2, True, 2, 31, True),
    (128, 64, 0, False, 2, 50, False),     # fp16 input, no LN
    (32, 96, 0, True, 1, 5, True),
    (192, 1024, 1, True, 2, 16401, False),  # k_pw_wide (>= 32 768 px at K = 192): ragged last group of each tile, 16 iterations of 2 row blocks
    (192, 520, 2, True, 3, 11000, True),    # k_pw_wide: 17 row blocks (odd: the last iteration has one), n_store not a multiple of 32, bias
    (192, 576, 0, False, 1, 33000, False),  # k_pw_wide: fp16 input, partial last workgroup
    (384, 1152, 1, True, 2, 8200, True),    # k_pw_wide at K = 384 (>= 16 384 px): one row block per iteration, 36 blocks
])
def test_pw_kres(ctx, K, N, ln, in_f32, T, P, with_bias):
    g = torch.Generator().manual_seed(K * 7 + N)
    x = torch.randn(T * P, K, generator=g) * 1.7 + 0.3
    if not in_f32:
        x = r16(x)
    NB = up32(N) // 32
    W = torch.randn(N, K, generator=g) / K ** 0.5
    Wp = rowperm_pad(W, NB * 32)
    bias = torch.randn(NB * 32, generator=g) * 0.1 if with_bias else None
    ref = (r16(ln_core(x, ln)).double() @ W.double().t()).float()
    if with_bias:
        ref = ref + bias[:N]
    y = torch.full((T * P, N), float('nan'), dtype=torch.float16).cuda()
    xd = dev(x if in_f32 else x.half())
    Wd = dev(split16(Wp))
    bd = dev(bias) if with_bias else None
    rc = ctx._lib.ira_op_pw_kres(ctx.handle, ptr(xd), int(in_f32), ln, K, ptr(Wd), ptr(bd), ptr(y), N, N, NB, T, P)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(y.cpu().float(), ref, 2e-3)


@pytest.mark.parametrize('K0,K1,in_f32,N,T,P,resid,per_tile,with_bias', [
    (96, 0, False, 96, 2, 70, True, True, False),     # attention apply: per-tile M on V + residual
    (256, 0, False, 48, 2, 45, True, False, False),   # project_out + residual (:92,149), Hp=128*2
    (192, 192, True, 192, 1, 64, False, False, True),  # reduce_chan on cat (:260-261)
    (64, 0, False, 16, 3, 9, True, False, False),     # reduced
    (1024, 0, False, 384, 1, 33, True, False, False),  # latent project_out
    (48, 48, True, 48, 2, 31, False, False, False),
    (128, 0, True, 160, 2, 300, True, False, True),    # k_pw_gemm: 5 row blocks as 3 + 2 (clamped partial chunk), several workgroups per image, fp32 pixels
    (96, 0, False, 224, 1, 257, False, False, True),   # k_pw_gemm: 7 row blocks as 4 + 3, ragged last workgroup
    (512, 0, False, 192, 2, 16401, True, False, False),  # k_nres_wide (>= 32 768 px, N = 192): project_out + residual at level 3, ragged last group of each tile
    (192, 0, False, 192, 3, 11000, True, True, False),   # k_nres_wide: attention apply, one matrix per tile, tiles that are no multiple of 256 px
    (1024, 0, False, 384, 1, 33000, True, False, True),  # k_nres_wide: N = 384 as two halves, bias, 16 stages
    (256, 0, False, 184, 1, 32800, False, False, False), # k_nres_wide: N not a multiple of 32 (partial last row block), no residual
    (480, 0, False, 240, 1, 33001, True, False, True),   # k_nres_wide with eight row blocks, stages of three k-steps (SwinIR-L fc2), ragged
    (256, 0, False, 256, 2, 16384, True, False, True),   # eight row blocks, stages of four (SwinIR-L proj)
])
def test_pw_nres(ctx, K0, K1, in_f32, N, T, P, resid, per_tile, with_bias):
    g = torch.Generator().manual_seed(K0 + 3 * N + K1)
    K = K0 + K1
    NB = up32(N) // 32
    x0 = torch.randn(T * P, K0, generator=g)
    x1 = torch.randn(T * P, K1, generator=g) if K1 else None
    if not in_f32:
        x0 = r16(x0)
    nW = T if per_tile else 1
    W = torch.randn(nW, N, K, generator=g) / K ** 0.5
    Wp = torch.zeros(nW, NB * 32, K); Wp[:, :N] = W
    bias = torch.randn(NB * 32, generator=g) * 0.1 if with_bias else None
    res = torch.randn(T * P, N, generator=g) if resid else None
    xin = torch.cat([x0, x1], 1) if K1 else x0
    xin16 = r16(xin).view(T, P, K)
    ref = torch.einsum('tpk,tnk->tpn', xin16.double(), (W.expand(T, -1, -1) if not per_tile else W).double()).reshape(T * P, N).float()
    if with_bias:
        ref = ref + bias[:N]
    if resid:
        ref = ref + res
    y = dev(res.clone()) if resid else torch.full((T * P, N), float('nan')).cuda()
    x0d = dev(x0 if in_f32 else x0.half())
    x1d = dev(x1) if K1 else None
    Wd = dev(split16(Wp).transpose(0, 1))            # [nW][2][NB][K/16][64][8]
    bd = dev(bias) if with_bias else None
    rc = ctx._lib.ira_op_pw_nres(ctx.handle, ptr(x0d), K0, ptr(x1d), K1, int(in_f32), ptr(Wd),
                                 2 * NB * 32 * K if per_tile else 0, ptr(bd), ptr(y) if resid else None, ptr(y),
                                 N, NB, T, P)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    # hi + lo weight planes: the result must be at fp32-accumulation accuracy, far below one fp16 ulp of W
    # (2.4e-4 relative).  This also proves the lo plane (often fp16-subnormal) is not flushed by the MFMA.
    close(y.cpu(), ref, 2e-5, 2e-5)


def unfrag(m):
    """inverse of frag_major: [..., NB, KS, 64, 8] -> [..., NB*32, KS*16]."""
    from ira_amd.pack import ROWPERM
    *lead, nb, ks, _, _ = m.shape
    a = m.reshape(*lead, nb, ks, 2, 32, 8)                  # [.., nb, s, h, r, j]
    n = len(lead)
    a = a.permute(*range(n), n, n + 3, n + 1, n + 2, n + 4)  # [.., nb, r, s, h, j]
    out = torch.empty_like(a)
    out[..., torch.from_numpy(ROWPERM), :, :, :] = a        # row rowperm(r) <- lane r
    return out.reshape(*lead, nb * 32, ks * 16)


def px_minor(t_nchw):
    """(T,Cq,H,W) -> (T,H,WB,Cq,8) zero-padded along W."""
    T, Cq, H, W = t_nchw.shape
    WB = (W + 7) // 8
    pad = torch.zeros(T, Cq, H, WB * 8)
    pad[..., :W] = t_nchw
    return pad.view(T, Cq, H, WB, 8).permute(0, 2, 3, 1, 4).contiguous()


@pytest.mark.parametrize('C,T,H,W', [(16, 2, 5, 13), (48, 1, 8, 16), (96, 1, 3, 7)])
def test_dwconv_qkv(ctx, C, T, H, W):
    g = torch.Generator().manual_seed(C + H)
    x = r16(torch.randn(T, 3 * C, H, W, generator=g))
    w = torch.randn(3 * C, 1, 3, 3, generator=g) / 3
    ref = F.conv2d(x, w, padding=1, groups=3 * C)          # restormer_arch.py:106,114
    q_ref = px_minor(ref[:, :2 * C])
    v_ref = ref[:, 2 * C:].permute(0, 2, 3, 1).contiguous()
    xd = dev(x.permute(0, 2, 3, 1).half())
    wd = dev(w.view(3 * C, 9).t())
    WB = (W + 7) // 8
    qk = torch.full((T, H, WB, 2 * C, 8), float('nan'), dtype=torch.float16).cuda()
    v = torch.full((T, H, W, C), float('nan'), dtype=torch.float16).cuda()
    rc = ctx._lib.ira_op_dwconv_qkv(ctx.handle, ptr(xd), ptr(wd), None, ptr(qk), ptr(v), T, H, W, C)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(v.cpu().float(), v_ref, 2e-3)
    close(qk.cpu().float(), q_ref, 2e-3)


def _hi_plane_ctx():
    """A handle whose op entry points run with ira_set_weight_planes(1) (sticky per handle: not the shared fixture)."""
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.weights import seeded_state_dict
    cfg = reduced_config()
    return RestormerBackend(cfg, seeded_state_dict(cfg, 1), device=0, weight_planes=1)


@pytest.mark.parametrize('K,N,ln', [(256, 768, 2), (192, 544, 1)])
def test_pw_kres_hi_plane_only(ctx, K, N, ln):
    """ira_set_weight_planes(1): k_pw_wide with the hi plane alone = plain fp16 weights (SwinIR-L LN + qkv shape; 17 row blocks)."""
    g = torch.Generator().manual_seed(K + N)
    T, P = 1, 33000
    x = torch.randn(T * P, K, generator=g) * 1.3 + 0.2
    NB = up32(N) // 32
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(NB * 32, generator=g) * 0.1
    y = torch.full((T * P, N), float('nan'), dtype=torch.float16).cuda()
    xd, Wd, bd = dev(x), dev(split16(rowperm_pad(W, NB * 32))), dev(bias)
    with _hi_plane_ctx() as c1:
        rc = c1._lib.ira_op_pw_kres_ex(c1.handle, ptr(xd), 1, ln, K, ptr(Wd), ptr(bd), ptr(y), N, N, NB, T, P, C.c_float(1e-5), 0, 0)
        assert rc == 0, c1._lib.ira_last_error(c1.handle)
    ref16 = (r16(ln_core(x, ln)).double() @ r16(W).double().t()).float() + bias[:N]
    ref32 = (r16(ln_core(x, ln)).double() @ W.double().t()).float() + bias[:N]
    got = y.cpu().float()
    close(got, ref16, 2e-3)
    assert (got - ref16).abs().mean() < (got - ref32).abs().mean()          # it really is the fp16-weight product


@pytest.mark.parametrize('K,N', [(480, 240), (256, 256), (512, 192)])
def test_pw_nres_hi_plane_only(ctx, K, N):
    """ira_set_weight_planes(1): k_nres_wide with the hi plane alone (SwinIR-L fc2 / proj shapes and the six-block form)."""
    g = torch.Generator().manual_seed(K * 3 + N)
    T, P = 1, 33001
    NB = up32(N) // 32
    x = r16(torch.randn(T * P, K, generator=g))
    W = torch.randn(N, K, generator=g) / K ** 0.5
    Wp = torch.zeros(1, NB * 32, K); Wp[0, :N] = W
    res = torch.randn(T * P, N, generator=g)
    y = torch.full((T * P, N), float('nan')).cuda()
    xd, Wd, rd = dev(x.half()), dev(split16(rowperm_pad(Wp[0], NB * 32))), dev(res)
    with _hi_plane_ctx() as c1:
        rc = c1._lib.ira_op_pw_nres(c1.handle, ptr(xd), K, None, 0, 0, ptr(Wd), 0, None, ptr(rd), ptr(y), N, NB, T, P)
        assert rc == 0, c1._lib.ira_last_error(c1.handle)
    ref16 = (x.double() @ r16(W).double().t()).float() + res
    close(y.cpu(), ref16, 2e-5, 2e-5)


@pytest.mark.parametrize('planes,npx', [(2, 32768 + 77), (1, 40000)])
def test_swin_mlp_fused(ctx, planes, npx):
    """k_swin_mlp (network_swinir.py:274-277, Mlp :14-30): x += fc2(GELU(fc1(LN(x)))) with the hidden activations on chip, against
    (a) the two kernels it replaces through an fp16 hidden tensor and (b) a torch fp32 statement with the same rounding
    points (fp16 LayerNorm output, fp16 hidden).  Ragged last workgroup; 240 real of 256 channels as SwinIR-L."""
    g = torch.Generator().manual_seed(planes * 7 + 1)
    Cp, Cr, hid = 256, 240, 480
    x = torch.zeros(npx, Cp); x[:, :Cr] = torch.randn(npx, Cr, generator=g) * 1.5 + 0.3
    W1 = torch.zeros(hid, Cp); W1[:, :Cr] = torch.randn(hid, Cr, generator=g) / Cr ** 0.5
    W2 = torch.zeros(Cp, hid); W2[:Cr] = torch.randn(Cr, hid, generator=g) / hid ** 0.5
    b1 = torch.randn(hid, generator=g) * 0.1
    b2 = torch.zeros(Cp); b2[:Cr] = torch.randn(Cr, generator=g) * 0.1
    W1d, W2d, b1d, b2d = dev(split16(W1)), dev(split16(W2)), dev(b1), dev(b2)
    xa, xb = dev(x), dev(x)
    hbuf = torch.empty(npx, hid, dtype=torch.float16).cuda()
    be = ctx if planes == 2 else _hi_plane_ctx()
    try:
        L, h = be._lib, be.handle
        rc = L.ira_op_swin_mlp(h, ptr(xa), ptr(W1d), ptr(b1d), ptr(W2d), ptr(b2d), Cp, hid, npx, C.c_float(1e-5), Cr)
        assert rc == 0, L.ira_last_error(h)
        rc = L.ira_op_pw_kres_ex(h, ptr(xb), 1, 2, Cp, ptr(W1d), ptr(b1d), ptr(hbuf), hid, hid, hid // 32, 1, npx, C.c_float(1e-5), Cr, 1)
        assert rc == 0, L.ira_last_error(h)
        rc = L.ira_op_pw_nres(h, ptr(hbuf), hid, None, 0, 0, ptr(W2d), 0, ptr(b2d), ptr(xb), ptr(xb), Cp, Cp // 32, 1, npx)
        assert rc == 0, L.ira_last_error(h)
    finally:
        if planes == 1:
            be.close()
    # the fused kernel takes GELU from the Phi table of the fused GDFN kernel (|error| < 1.1e-6 |a|), the two-kernel path from the erf
    # polynomial: the fp16 hidden values differ by one unit in a few places, and with them the last bits of most outputs
    close(xa.cpu(), xb.cpu(), 2e-4, 2e-4)
    xr = x[:, :Cr]
    mu = xr.mean(-1, keepdim=True); var = xr.var(-1, keepdim=True, unbiased=False)
    n = r16((xr - mu) / torch.sqrt(var + 1e-5))
    Wa, Wb = (r16(W1), r16(W2)) if planes == 1 else (W1, W2)
    hd = r16(F.gelu((n.double() @ Wa[:, :Cr].double().t()).float() + b1))
    ref = x.clone(); ref[:, :Cr] += ((hd.double() @ Wb[:Cr].double().t()).float() + b2[:Cr])
    close(xa.cpu()[:, :Cr], ref[:, :Cr], 1e-3, 1e-3)        # fp16 rounding of the hidden values can flip by one unit next to ties


@pytest.mark.parametrize('planes,npx', [(2, 32768 + 45), (1, 36001)])
def test_swin_proj_mlp_fused(ctx, planes, npx):
    """k_swin_mlp<PROJ>: the second half of a Swin block (network_swinir.py:263-277) - attention projection + residual, then the MLP
    half-block - in one kernel, against ira_op_pw_nres followed by ira_op_swin_mlp on the same buffers: bit for bit (the projection keeps
    k_nres_wide's arithmetic, the LayerNorm sums keep the order of the load-based prologue)."""
    g = torch.Generator().manual_seed(planes * 11 + 3)
    Cp, Cr, hid = 256, 240, 480
    x = torch.zeros(npx, Cp); x[:, :Cr] = torch.randn(npx, Cr, generator=g) * 1.5 + 0.3
    att = r16(torch.randn(npx, Cp, generator=g))
    Wp = torch.zeros(Cp, Cp); Wp[:Cr] = torch.randn(Cr, Cp, generator=g) / Cp ** 0.5
    bp = torch.zeros(Cp); bp[:Cr] = torch.randn(Cr, generator=g) * 0.1
    W1 = torch.zeros(hid, Cp); W1[:, :Cr] = torch.randn(hid, Cr, generator=g) / Cr ** 0.5
    W2 = torch.zeros(Cp, hid); W2[:Cr] = torch.randn(Cr, hid, generator=g) / hid ** 0.5
    b1 = torch.randn(hid, generator=g) * 0.1
    b2 = torch.zeros(Cp); b2[:Cr] = torch.randn(Cr, generator=g) * 0.1
    Wpd, W1d, W2d, bpd, b1d, b2d, ad = dev(split16(Wp)), dev(split16(W1)), dev(split16(W2)), dev(bp), dev(b1), dev(b2), dev(att.half())
    xa, xb = dev(x), dev(x)
    be = ctx if planes == 2 else _hi_plane_ctx()
    try:
        L, h = be._lib, be.handle
        rc = L.ira_op_swin_proj_mlp(h, ptr(xa), ptr(ad), ptr(Wpd), ptr(bpd), ptr(W1d), ptr(b1d), ptr(W2d), ptr(b2d), Cp, hid, npx, C.c_float(1e-5), Cr)
        assert rc == 0, L.ira_last_error(h)
        rc = L.ira_op_pw_nres(h, ptr(ad), Cp, None, 0, 0, ptr(Wpd), 0, ptr(bpd), ptr(xb), ptr(xb), Cp, Cp // 32, 1, npx)
        assert rc == 0, L.ira_last_error(h)
        x1 = xb.clone()
        rc = L.ira_op_swin_mlp(h, ptr(xb), ptr(W1d), ptr(b1d), ptr(W2d), ptr(b2d), Cp, hid, npx, C.c_float(1e-5), Cr)
        assert rc == 0, L.ira_last_error(h)
    finally:
        if planes == 1:
            be.close()
    assert torch.equal(xa, xb)
    Wq = r16(Wp) if planes == 1 else Wp
    ref1 = x.clone(); ref1[:, :Cr] += (att.double() @ Wq[:Cr].double().t()).float() + bp[:Cr]
    close(x1.cpu()[:, :Cr], ref1[:, :Cr], 2e-5, 2e-5)          # the projection step of the two-kernel path itself


@pytest.mark.parametrize('Hp,T,H,W,with_bias', [(64, 2, 6, 9, False), (128, 1, 4, 24, False), (32, 1, 2, 3, False),
                                                (1024, 1, 5, 10, True),      # k_dwconv_gate_row: two chunks of 64 channel octets, rows not a multiple of 4
                                                (512, 2, 9, 37, True)])       # ragged last 4-pixel step, several row segments per band
def test_dwconv_gate(ctx, Hp, T, H, W, with_bias):
    g = torch.Generator().manual_seed(Hp + W)
    x = r16(torch.randn(T, 2 * Hp, H, W, generator=g))
    w = torch.randn(2 * Hp, 1, 3, 3, generator=g) / 3
    b = torch.randn(2 * Hp, generator=g) * 0.2 if with_bias else None
    d = F.conv2d(x, w, b, padding=1, groups=2 * Hp)         # restormer_arch.py:84,90
    ref = (F.gelu(d[:, :Hp]) * d[:, Hp:]).permute(0, 2, 3, 1).contiguous()   # :91
    xd = dev(x.permute(0, 2, 3, 1).half())
    wd = dev(w.view(2 * Hp, 9).t())
    bd = dev(b) if with_bias else None
    out = torch.full((T, H, W, Hp), float('nan'), dtype=torch.float16).cuda()
    rc = ctx._lib.ira_op_dwconv_gate(ctx.handle, ptr(xd), ptr(wd), ptr(bd), ptr(out), T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(out.cpu().float(), ref, 2e-3)


@pytest.mark.parametrize('C,heads,T,H,W,S', [(96, 1, 2, 6, 20, 3), (96, 2, 1, 9, 16, 2), (32, 2, 2, 4, 5, 1),
                                             (32, 1, 1, 7, 9, 4), (384, 8, 1, 3, 8, 2)])
def test_gram(ctx, C, heads, T, H, W, S):
    g = torch.Generator().manual_seed(C + heads + W)
    c = C // heads
    qk = r16(torch.randn(T, 2 * C, H, W, generator=g))
    qkd = dev(px_minor(qk).half())
    nblk = H * ((W + 7) // 8)
    rec = c * c + 2 * c
    part = torch.full((T, S, heads, rec), float('nan')).cuda()
    rc = ctx._lib.ira_op_gram(ctx.handle, ptr(qkd), ptr(part), T, nblk, C, heads, S)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    got = part.cpu().sum(1)                                   # (T, heads, rec)
    q = qk[:, :C].reshape(T, heads, c, H * W)
    k = qk[:, C:].reshape(T, heads, c, H * W)
    gram = q @ k.transpose(-1, -2)                            # restormer_arch.py:124
    close(got[..., :c * c].reshape(T, heads, c, c), gram, 1e-3)
    close(got[..., c * c:c * c + c], q.pow(2).sum(-1), 1e-3)
    close(got[..., c * c + c:], k.pow(2).sum(-1), 1e-3)


@pytest.mark.parametrize('C,heads,T,S', [(96, 1, 2, 3), (96, 2, 1, 1), (48, 1, 2, 2), (32, 2, 1, 2), (128, 8, 2, 1)])
def test_attn_mix(ctx, C, heads, T, S):
    g = torch.Generator().manual_seed(C * heads + S)
    c = C // heads
    n = 500
    q = torch.randn(T, heads, c, n, generator=g)
    k = torch.randn(T, heads, c, n, generator=g) + 0.5 * q
    gram = q @ k.transpose(-1, -2)
    nq, nk = q.pow(2).sum(-1), k.pow(2).sum(-1)
    full = torch.cat([gram.reshape(T, heads, c * c), nq, nk], -1)       # (T, heads, rec)
    wts = torch.rand(S, generator=g) + 0.1
    wts = wts / wts.sum()
    part = torch.stack([full * wts[s] for s in range(S)], 1).contiguous()   # (T,S,heads,rec)
    wo = torch.randn(C, C, generator=g) / C ** 0.5
    temp = torch.rand(heads, generator=g) * 5 + 1
    # reference: restormer_arch.py:121-131 re-associated
    logits = gram / (nq.sqrt().clamp_min(1e-12).unsqueeze(-1) * nk.sqrt().clamp_min(1e-12).unsqueeze(-2))
    attn = (logits * temp.view(1, heads, 1, 1)).softmax(-1)
    ref = torch.einsum('ohi,thij->tohj', wo.view(C, heads, c), attn).reshape(T, C, C)
    NB = up32(C) // 32
    M = torch.zeros(T, 2, NB, C // 16, 64, 8, dtype=torch.float16).cuda()
    pd, wod, td = dev(part), dev(wo), dev(temp)     # keep the device tensors alive across the call
    rc = ctx._lib.ira_op_attn_mix(ctx.handle, ptr(pd), ptr(wod), ptr(td), ptr(M), T, C, heads, S, NB)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    Mc = unfrag(M.cpu().float())                         # [T][2][NB*32][C]
    close(Mc[:, 0, :C], ref, 2e-3)                       # hi plane alone: fp16 accuracy
    close(Mc[:, 0, :C] + Mc[:, 1, :C], ref, 2e-5, 2e-5)  # hi + lo: fp32 accuracy


def conv3_pack(w):
    n, k = w.shape[:2]
    out = torch.zeros(9, up32(n), k)
    out[:, :n] = w.reshape(n, k, 9).permute(2, 0, 1)
    return split16(out)                                  # [2][9][rows][K]


@pytest.mark.parametrize('mode,K,N,T,H,W', [
    (0, 32, 40, 1, 5, 7),
    (1, 48, 24, 2, 6, 10),     # Downsample: conv + PixelUnshuffle (restormer_arch.py:171-179)
    (1, 192, 96, 1, 4, 4),
    (2, 32, 64, 2, 3, 5),      # Upsample: conv + PixelShuffle (:181-189)
    (2, 384, 768, 1, 2, 3),
    (2, 96, 192, 2, 11, 21),   # six row blocks through the LDS-staged PixelShuffle epilogue, ragged tiles in both directions
    (1, 96, 48, 2, 10, 22),    # Downsample with ragged 8 x 16 tiles
    (2, 64, 256, 1, 4, 4),     # NB=8: two n-chunks, the second partial
    (3, 96, 3, 2, 8, 9),       # output conv + input skip (:281)
    (3, 32, 1, 1, 5, 5),
])
def test_conv3x3(ctx, mode, K, N, T, H, W):
    g = torch.Generator().manual_seed(mode * 100 + K + N)
    x = torch.randn(T, K, H, W, generator=g)
    w = torch.randn(N, K, 3, 3, generator=g) / (9 * K) ** 0.5
    y = F.conv2d(x.double(), w.double(), padding=1).float()      # input and weights are both hi/lo split
    skip = None
    if mode == 1:
        y = F.pixel_unshuffle(y, 2)
    elif mode == 2:
        y = F.pixel_shuffle(y, 2)
    elif mode == 3:
        skip = torch.randn(T, N, H, W, generator=g)
        y = y + skip
    ref = y.permute(0, 2, 3, 1).contiguous()
    out = torch.full(ref.shape, float('nan')).cuda()
    xd = dev(x.permute(0, 2, 3, 1))
    sd = dev(skip.permute(0, 2, 3, 1)) if skip is not None else None
    NB = up32(N) // 32
    wd = dev(conv3_pack(w))
    rc = ctx._lib.ira_op_conv3x3(ctx.handle, ptr(xd), ptr(wd), None, ptr(sd), ptr(out), mode, T, H, W, K, N, NB)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(out.cpu(), ref, 2e-5, 2e-5)


@pytest.mark.parametrize('ic,Cc,T,H,W', [(3, 48, 2, 5, 9), (1, 16, 1, 8, 8)])
def test_embed(ctx, ic, Cc, T, H, W):
    g = torch.Generator().manual_seed(ic + Cc)
    x = torch.rand(T, ic, H, W, generator=g)
    w = torch.randn(Cc, ic, 3, 3, generator=g) / (9 * ic) ** 0.5
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1).contiguous()      # restormer_arch.py:156-165
    out = torch.full(ref.shape, float('nan')).cuda()
    xd, wd = dev(x.permute(0, 2, 3, 1)), dev(w)
    rc = ctx._lib.ira_op_embed(ctx.handle, ptr(xd), ptr(wd), None, ptr(out), T, H, W, ic, Cc)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    close(out.cpu(), ref, 1e-5, 1e-6)


def test_concat2(ctx):
    a = torch.randn(37, 48)
    b = torch.randn(37, 16)
    out = torch.zeros(37, 64).cuda()
    ad, bd = dev(a), dev(b)
    rc = ctx._lib.ira_op_concat2(ctx.handle, ptr(ad), 48, ptr(bd), 16, ptr(out), 37)
    assert rc == 0
    assert torch.equal(out.cpu(), torch.cat([a, b], 1))     # restormer_arch.py:270


@pytest.mark.parametrize('C,hid,ln,T,H,W,bias', [
    (48, 127, 1, 2, 11, 40, False),     # level-1 encoder block, ragged patch edges (H % 8, W % 32 != 0)
    (96, 255, 1, 1, 16, 64, False),     # decoder level 1 / refinement / level 2
    (96, 255, 2, 2, 9, 33, True),       # WithBias LN + conv biases
    (48, 127, 2, 1, 8, 32, False),
])
def test_ffn_fused(ctx, C, hid, ln, T, H, W, bias):
    """Whole GDFN block + residual in one kernel (restormer_arch.py:88-93,149) against the fp32 torch ops."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    g = torch.Generator().manual_seed(C + hid + ln)
    Hp = up32(hid)
    x = torch.randn(T, C, H, W, generator=g) * 1.5
    sd = {
        'b.norm1.body.weight': torch.ones(C), 'b.norm2.body.weight': 1 + 0.1 * torch.randn(C, generator=g),
        'b.attn.temperature': torch.ones(1, 1, 1), 'b.attn.qkv.weight': torch.zeros(3 * C, C, 1, 1),
        'b.attn.qkv_dwconv.weight': torch.zeros(3 * C, 1, 3, 3), 'b.attn.project_out.weight': torch.zeros(C, C, 1, 1),
        'b.ffn.project_in.weight': torch.randn(2 * hid, C, 1, 1, generator=g) / C ** 0.5,
        'b.ffn.dwconv.weight': torch.randn(2 * hid, 1, 3, 3, generator=g) / 3,
        'b.ffn.project_out.weight': torch.randn(C, hid, 1, 1, generator=g) / hid ** 0.5,
    }
    if ln == 2:
        sd['b.norm1.body.bias'] = torch.zeros(C)
        sd['b.norm2.body.bias'] = 0.1 * torch.randn(C, generator=g)
    if bias:
        for k, n in [('attn.qkv', 3 * C), ('attn.qkv_dwconv', 3 * C), ('attn.project_out', C), ('ffn.project_in', 2 * hid),
                     ('ffn.dwconv', 2 * hid), ('ffn.project_out', C)]:
            sd[f'b.{k}.bias'] = 0.1 * torch.randn(n, generator=g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree' if ln == 1 else 'WithBias', bias=bias)
    cfg.hidden = lambda d: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    # reference
    from oracle import restormer_oracle as O
    ln_type = 'BiasFree' if ln == 1 else 'WithBias'
    xn = O.layer_norm(x, sd['b.norm2.body.weight'], sd.get('b.norm2.body.bias'), ln_type)
    ref = (x + O.feed_forward(xn, sd, 'b.ffn')).permute(0, 2, 3, 1).contiguous()
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v))) for k, v in arrs.items()}
    xd = dev(x.permute(0, 2, 3, 1))
    out = torch.full(ref.shape, float('nan')).cuda()
    rc = ctx._lib.ira_op_ffn_fused(ctx.handle, ptr(xd), ptr(out), ln, C, ptr(d['b.pin_w']), ptr(d.get('b.pin_b')),
                                   ptr(d['b.dw2_w']), ptr(d.get('b.dw2_b')), ptr(d['b.pout_w']), ptr(d.get('b.pout_b')),
                                   T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    got = out.cpu()
    assert torch.isfinite(got).all()
    # fp16 LN output / hidden / gated values inside: a few 1e-3 relative to the branch magnitude
    err = (got - ref).abs().max().item()
    scale = (ref - x.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 4e-3 * scale + 1e-4, (err, scale)


def _ffn_block_sd(C, hid, ln, bias, g):
    sd = {
        'b.norm1.body.weight': torch.ones(C), 'b.norm2.body.weight': 1 + 0.1 * torch.randn(C, generator=g),
        'b.attn.temperature': torch.ones(1, 1, 1), 'b.attn.qkv.weight': torch.zeros(3 * C, C, 1, 1),
        'b.attn.qkv_dwconv.weight': torch.zeros(3 * C, 1, 3, 3), 'b.attn.project_out.weight': torch.zeros(C, C, 1, 1),
        'b.ffn.project_in.weight': torch.randn(2 * hid, C, 1, 1, generator=g) / C ** 0.5,
        'b.ffn.dwconv.weight': torch.randn(2 * hid, 1, 3, 3, generator=g) / 3,
        'b.ffn.project_out.weight': torch.randn(C, hid, 1, 1, generator=g) / hid ** 0.5,
    }
    if ln == 2:
        sd['b.norm1.body.bias'] = torch.zeros(C)
        sd['b.norm2.body.bias'] = 0.1 * torch.randn(C, generator=g)
    if bias:
        for k, n in [('attn.qkv', 3 * C), ('attn.qkv_dwconv', 3 * C), ('attn.project_out', C), ('ffn.project_in', 2 * hid),
                     ('ffn.dwconv', 2 * hid), ('ffn.project_out', C)]:
            sd[f'b.{k}.bias'] = 0.1 * torch.randn(n, generator=g)
    return sd


@pytest.mark.parametrize('C,hid,ln,T,H,W,bias,fold', [
    (96, 255, 1, 2, 16, 64, False, False),     # level-1 decoder shape (Hf = 255 -> 256), whole patches
    (48, 127, 1, 1, 8, 32, False, False),
    (96, 255, 1, 1, 13, 45, False, False),     # ragged: patches hang over the right and bottom image edges
    (96, 255, 2, 2, 9, 33, True, False),       # WithBias LN + conv biases
    (48, 127, 2, 1, 19, 70, False, False),
    (96, 255, 1, 2, 16, 64, False, True),      # + the attention apply x1 = x + M_t . v folded into the prologue (per-tile M)
    (48, 127, 1, 3, 11, 40, False, True),
    (96, 255, 2, 2, 9, 37, True, True),        # fold with the project_out bias of the attention, WithBias LN
    (48, 95, 2, 1, 24, 31, True, True),        # hidden width 95 -> 96: an odd number of 32-channel blocks, padded gate channels
])
def test_ffn2(ctx, C, hid, ln, T, H, W, bias, fold):
    """Second-generation fused block (csrc/ira_ffn2.hip): [x1 = x + M_t . v (+ b_o)] + GDFN + residual in one kernel
    (restormer_arch.py:127-131 as a per-tile 1x1, :88-93, :146-150) against the fp32 torch ops."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    from oracle import restormer_oracle as O
    g = torch.Generator().manual_seed(C + hid + ln + H)
    Hp = up32(hid)
    x = torch.randn(T, C, H, W, generator=g) * 1.5
    sd = _ffn_block_sd(C, hid, ln, bias, g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree' if ln == 1 else 'WithBias', bias=bias)
    cfg.hidden = lambda d: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    ln_type = 'BiasFree' if ln == 1 else 'WithBias'
    x1 = x
    vd = Md = wob = None
    NBc = up32(C) // 32
    if fold:
        v = r16(torch.randn(T, C, H, W, generator=g))
        M = torch.randn(T, C, C, generator=g) / C ** 0.5
        x1 = x + torch.einsum('toc,tchw->tohw', M, v)
        if bias:
            x1 = x1 + sd['b.attn.project_out.bias'].view(1, -1, 1, 1)
            wob = dev(torch.from_numpy(arrs['b.wo_b']))
        Mp = torch.zeros(T, NBc * 32, C)
        Mp[:, :C] = M
        Md = dev(split16(Mp).permute(1, 0, 2, 3, 4, 5))                 # [T][hi, lo][NB][C/16][64][8]
        vd = dev(v.permute(0, 2, 3, 1).half())
    xn = O.layer_norm(x1, sd['b.norm2.body.weight'], sd.get('b.norm2.body.bias'), ln_type)
    ref = (x1 + O.feed_forward(xn, sd, 'b.ffn')).permute(0, 2, 3, 1).contiguous()
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v_))) for k, v_ in arrs.items()}
    assert d['b.dw2_u'].shape == (Hp // 16, 2, 18, 8)
    xd = dev(x.permute(0, 2, 3, 1))
    out = torch.full(ref.shape, float('nan')).cuda()
    rc = ctx._lib.ira_op_ffn2(ctx.handle, ptr(xd), ptr(out), ptr(vd), ptr(Md), 2 * NBc * 32 * C, ptr(wob), ln, C, ptr(d['b.pin_w']),
                              ptr(d.get('b.pin_b')), ptr(d['b.dw2_u']), ptr(d.get('b.dw2_ub')), ptr(d['b.pout_w']), ptr(d.get('b.pout_b')),
                              T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    got = out.cpu()
    assert torch.isfinite(got).all(), f"{int((~torch.isfinite(got)).sum())} non-finite values"
    # fp16 LN output / hidden / gated values inside: a few 1e-3 relative to the branch magnitude
    err = (got - ref).abs()
    scale = (ref - x.permute(0, 2, 3, 1)).abs().max().item()
    assert err.max().item() <= 4e-3 * scale + 1e-4, (err.max().item(), scale, np.unravel_index(int(err.argmax()), err.shape))


def test_ffn2_matches_generation_1(ctx):
    """The same block through ira_op_ffn_fused (generation 1, A&S erf) and ira_op_ffn2 (Phi table): equal to table accuracy."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    C, hid, T, H, W = 96, 255, 2, 24, 64
    g = torch.Generator().manual_seed(7)
    Hp = up32(hid)
    x = torch.randn(T, H, W, C, generator=g) * 1.5
    sd = _ffn_block_sd(C, hid, 1, False, g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree')
    cfg.hidden = lambda d: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v_))) for k, v_ in arrs.items()}
    xd = dev(x)
    o1 = torch.full(x.shape, float('nan')).cuda(); o2 = torch.full(x.shape, float('nan')).cuda()
    rc = ctx._lib.ira_op_ffn_fused(ctx.handle, ptr(xd), ptr(o1), 1, C, ptr(d['b.pin_w']), None, ptr(d['b.dw2_w']), None, ptr(d['b.pout_w']), None, T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    rc = ctx._lib.ira_op_ffn2(ctx.handle, ptr(xd), ptr(o2), None, None, 0, None, 1, C, ptr(d['b.pin_w']), None, ptr(d['b.dw2_u']), None, ptr(d['b.pout_w']), None, T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    dlt = (o1 - o2).abs().max().item()
    br = (o1 - xd).abs().max().item()
    print(f"generation 1 vs 2: max |delta| {dlt:.3e} on a branch of magnitude {br:.3f}")
    # same rounding points, but the fp32 sums feeding them are formed in another order (and GELU differs by 1e-6): a hidden value next to
    # a rounding boundary lands on the other fp16 neighbour (2^-11 relative), which is what the two outputs differ by
    assert dlt <= 5e-4 * br + 1e-6


@pytest.mark.parametrize('lo_mask', [0, 1])
@pytest.mark.parametrize('ln,bias,fold,T,H,W', [(1, False, True, 2, 24, 64), (2, True, True, 1, 19, 45), (1, False, False, 1, 16, 96), (2, True, False, 2, 8, 32)])
def test_ffn2_plain_fp16_forms(ctx, lo_mask, ln, bias, fold, T, H, W):
    """The plain-fp16 forms of k_ffn2 (lo_mask 0: neither 1x1 has a lo plane, 1: project_in alone has one; the schedule runs them only
    under the A/B switch IRA_FFN_LO_STAGES): on weights whose dropped lo planes are ZERO - what the handle makes of its blob
    then - they must give the two-plane kernel's output bit for bit (a tile's result may not depend on which form ran), and they never
    read the dropped plane (poisoned here with NaN)."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    C, hid = 96, 255
    g = torch.Generator().manual_seed(31 * lo_mask + ln + H)
    Hp = up32(hid)
    x = torch.randn(T, H, W, C, generator=g) * 1.5
    sd = _ffn_block_sd(C, hid, ln, bias, g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree' if ln == 1 else 'WithBias', bias=bias)
    cfg.hidden = lambda d: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    NBc = up32(C) // 32
    vd = Md = wob = None
    if fold:
        Mp = torch.zeros(T, NBc * 32, C)
        Mp[:, :C] = torch.randn(T, C, C, generator=g) / C ** 0.5
        Md = dev(split16(Mp).permute(1, 0, 2, 3, 4, 5))
        vd = dev(torch.randn(T, H, W, C, generator=g).half())
        if bias:
            wob = dev(torch.from_numpy(arrs['b.wo_b']))

    def planes(name, lo_value):
        w = torch.from_numpy(np.ascontiguousarray(arrs[name])).clone().view(torch.int16).reshape(-1)
        assert w.numel() % 2 == 0
        if lo_value is not None:
            w[w.numel() // 2:] = lo_value                 # the lo plane is the second half of the packed array (pack.py::split16 layout)
        return dev(w)
    drop_in, drop_out = not (lo_mask & 1), not (lo_mask & 2)
    NAN16 = 0x7e00
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v_))) for k, v_ in arrs.items()}
    win0, wout0 = planes('b.pin_w', 0 if drop_in else None), planes('b.pout_w', 0 if drop_out else None)
    winN, woutN = planes('b.pin_w', NAN16 if drop_in else None), planes('b.pout_w', NAN16 if drop_out else None)
    xd = dev(x)
    o2 = torch.full(x.shape, float('nan')).cuda(); o1 = torch.full(x.shape, float('nan')).cuda()
    tail = (ptr(d.get('b.pin_b')), ptr(d['b.dw2_u']), ptr(d.get('b.dw2_ub')))
    rc = ctx._lib.ira_op_ffn2(ctx.handle, ptr(xd), ptr(o2), ptr(vd), ptr(Md), 2 * NBc * 32 * C, ptr(wob), ln, C, ptr(win0), *tail, ptr(wout0),
                              ptr(d.get('b.pout_b')), T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    rc = ctx._lib.ira_op_ffn2_planes(ctx.handle, lo_mask, ptr(xd), ptr(o1), ptr(vd), ptr(Md), 2 * NBc * 32 * C, ptr(wob), ln, C, ptr(winN), *tail,
                                     ptr(woutN), ptr(d.get('b.pout_b')), T, H, W, Hp)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    assert torch.isfinite(o1).all(), "the plain-fp16 form read a lo plane it should not"
    assert torch.equal(o1, o2), f"{int((o1 != o2).sum())} outputs differ, max {float((o1 - o2).abs().max()):.3e}"


def test_ffn2_plain_fp16_forms_only_where_built(ctx):
    """lo masks without an instantiation are an argument error, not a silent two-plane run."""
    z = torch.zeros(4096, device='cuda')
    for C, mask in [(48, 0), (48, 1), (96, 2), (96, 5)]:
        rc = ctx._lib.ira_op_ffn2_planes(ctx.handle, mask, ptr(z), ptr(z), None, None, 0, None, 1, C, ptr(z), None, ptr(z), None, ptr(z), None, 1, 8, 32, 32)
        assert rc != 0 and 'lo_mask' in ctx._lib.ira_last_error(ctx.handle).decode()


@pytest.mark.parametrize('C,ln,T,H,W,bias', [
    (48, 1, 2, 11, 40, False),      # ragged patch edges; W % 8 == 0
    (96, 1, 1, 16, 64, False),
    (96, 2, 2, 9, 37, True),        # WithBias LN + biases; W % 8 = 5: zero-padded pixel-minor rows
    (48, 2, 1, 8, 33, False),       # W % 8 = 1
    (96, 1, 3, 40, 100, False),     # 60 patches, interior ones (the compile-time inside path of generation 3) and every kind of border
    (48, 1, 2, 24, 96, True),
])
@pytest.mark.parametrize('gen', [1, 3])
def test_qkv_fused(ctx, C, ln, T, H, W, bias, gen):
    """LN + qkv 1x1 + depthwise 3x3 in one kernel (restormer_arch.py:105-106,114) against the fp32 torch ops; both kernel generations
    (1: ira_qkv_fused.hip, 3: ira_qkv3.hip, the one the network schedule runs)."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    from oracle import restormer_oracle as O
    g = torch.Generator().manual_seed(C + ln + W)
    hid = 32
    x = torch.randn(T, C, H, W, generator=g) * 1.5
    sd = {
        'b.norm1.body.weight': 1 + 0.1 * torch.randn(C, generator=g), 'b.norm2.body.weight': torch.ones(C),
        'b.attn.temperature': torch.ones(1, 1, 1), 'b.attn.qkv.weight': torch.randn(3 * C, C, 1, 1, generator=g) / C ** 0.5,
        'b.attn.qkv_dwconv.weight': torch.randn(3 * C, 1, 3, 3, generator=g) / 3,
        'b.attn.project_out.weight': torch.zeros(C, C, 1, 1),
        'b.ffn.project_in.weight': torch.zeros(2 * hid, C, 1, 1), 'b.ffn.dwconv.weight': torch.zeros(2 * hid, 1, 3, 3),
        'b.ffn.project_out.weight': torch.zeros(C, hid, 1, 1),
    }
    if ln == 2:
        sd['b.norm1.body.bias'] = 0.1 * torch.randn(C, generator=g)
        sd['b.norm2.body.bias'] = torch.zeros(C)
    if bias:
        for k, n in [('attn.qkv', 3 * C), ('attn.qkv_dwconv', 3 * C), ('attn.project_out', C), ('ffn.project_in', 2 * hid),
                     ('ffn.dwconv', 2 * hid), ('ffn.project_out', C)]:
            sd[f'b.{k}.bias'] = 0.1 * torch.randn(n, generator=g)
    ln_type = 'BiasFree' if ln == 1 else 'WithBias'
    cfg = RestormerConfig(dim=C, LayerNorm_type=ln_type, bias=bias)
    cfg.hidden = lambda d: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    xn = O.layer_norm(x, sd['b.norm1.body.weight'], sd.get('b.norm1.body.bias'), ln_type)
    qkv = F.conv2d(F.conv2d(xn, sd['b.attn.qkv.weight'], sd.get('b.attn.qkv.bias')), sd['b.attn.qkv_dwconv.weight'],
                   sd.get('b.attn.qkv_dwconv.bias'), padding=1, groups=3 * C)
    q_ref = px_minor(qkv[:, :2 * C])
    v_ref = qkv[:, 2 * C:].permute(0, 2, 3, 1).contiguous()
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v))) for k, v in arrs.items()}
    xd = dev(x.permute(0, 2, 3, 1))
    WB = (W + 7) // 8
    qk = torch.full((T, H, WB, 2 * C, 8), float('nan'), dtype=torch.float16).cuda()
    v = torch.full((T, H, W, C), float('nan'), dtype=torch.float16).cuda()
    rc = ctx._lib.ira_op_qkv_fused_gen(ctx.handle, gen, ptr(xd), ln, C, ptr(d['b.qkv_w']), ptr(d.get('b.qkv_b')), ptr(d['b.dw1_w']),
                                       ptr(d.get('b.dw1_b')), ptr(qk), ptr(v), T, H, W)
    assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
    # fp16 LN output and fp16 1x1 output inside: a few fp16 ulps of the intermediate magnitudes
    close(v.cpu().float(), v_ref, 4e-3, 3e-3)
    close(qk.cpu().float(), q_ref, 4e-3, 3e-3)


@pytest.mark.parametrize('C,ln,T,H,W,bias', [
    (96, 1, 4, 96, 224, False),     # 336 patches: more than one per CU (several workgroups per CU in turn), all interior / border kinds
    (48, 2, 5, 72, 230, True),      # 360 patches, ragged right edge (W % 32 = 6, W % 8 = 6), WithBias LN + biases
    (96, 2, 1, 17, 41, True),
    (48, 1, 1, 512, 512, False),    # one benchmark tile at level 1 (1024 patches)
])
@pytest.mark.parametrize('GEN_NEW', [3])
def test_qkv3_equals_generation1(ctx, C, ln, T, H, W, bias, GEN_NEW):
    """The third-generation kernel has the same rounding points as generation 1 (fp16 LayerNorm output, fp16 1x1 output, fp32
    depthwise sums, fp16 results) and forms the same products; the only difference is that its 1x1 is the TRANSPOSED MFMA (pixels as rows),
    whose fp32 sums are formed in another order.  Measured: about 6e-6 of the hidden values land on the other fp16 neighbour (each
    reaches the 3 x 3 outputs around it); everything else is bit-identical.  Asserted: differing outputs < 1e-3 of all, each by at most one
    fp16 ulp of the 1x1 output's magnitude scaled by the depthwise weight, and the new generation is deterministic from launch to launch."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    g = torch.Generator().manual_seed(7 * C + ln + W)
    hid = 32
    sd = {
        'b.norm1.body.weight': 1 + 0.1 * torch.randn(C, generator=g), 'b.norm2.body.weight': torch.ones(C),
        'b.attn.temperature': torch.ones(1, 1, 1), 'b.attn.qkv.weight': torch.randn(3 * C, C, 1, 1, generator=g) / C ** 0.5,
        'b.attn.qkv_dwconv.weight': torch.randn(3 * C, 1, 3, 3, generator=g) / 3,
        'b.attn.project_out.weight': torch.zeros(C, C, 1, 1),
        'b.ffn.project_in.weight': torch.zeros(2 * hid, C, 1, 1), 'b.ffn.dwconv.weight': torch.zeros(2 * hid, 1, 3, 3),
        'b.ffn.project_out.weight': torch.zeros(C, hid, 1, 1),
    }
    if ln == 2:
        sd['b.norm1.body.bias'] = 0.1 * torch.randn(C, generator=g)
        sd['b.norm2.body.bias'] = torch.zeros(C)
    if bias:
        for k, n in [('attn.qkv', 3 * C), ('attn.qkv_dwconv', 3 * C), ('attn.project_out', C), ('ffn.project_in', 2 * hid),
                     ('ffn.dwconv', 2 * hid), ('ffn.project_out', C)]:
            sd[f'b.{k}.bias'] = 0.1 * torch.randn(n, generator=g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree' if ln == 1 else 'WithBias', bias=bias)
    cfg.hidden = lambda d_: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v_))) for k, v_ in arrs.items()}
    xd = (torch.randn(T, H, W, C, generator=g) * 1.5).cuda()
    WB = (W + 7) // 8
    outs = []
    for gen in (1, GEN_NEW, GEN_NEW):                  # the new generation twice, into fresh buffers: a persistent kernel must not depend on what the last launch left
        qk = torch.full((T, H, WB, 2 * C, 8), float('nan'), dtype=torch.float16).cuda()
        v = torch.full((T, H, W, C), float('nan'), dtype=torch.float16).cuda()
        rc = ctx._lib.ira_op_qkv_fused_gen(ctx.handle, gen, ptr(xd), ln, C, ptr(d['b.qkv_w']), ptr(d.get('b.qkv_b')), ptr(d['b.dw1_w']),
                                           ptr(d.get('b.dw1_b')), ptr(qk), ptr(v), T, H, W)
        assert rc == 0, ctx._lib.ira_last_error(ctx.handle)
        torch.cuda.synchronize()
        outs.append((qk, v))
    for k, name in ((0, 'q,k'), (1, 'v')):
        a, b, b2 = outs[0][k], outs[1][k], outs[2][k]
        assert torch.isfinite(a.float()).all() and torch.isfinite(b.float()).all()
        assert torch.equal(b.view(torch.int16), b2.view(torch.int16)), f"{name}: generation {GEN_NEW} differs from launch to launch"
        df = (a.float() - b.float()).abs()
        ndiff = int((df > 0).sum())
        # one fp16 ulp of a hidden value of magnitude < 8 is 2^-8 at most; a depthwise weight is < 2 here
        print(f"{name}: {ndiff} of {a.numel()} elements differ, max |delta| {float(df.max()):.3e}")
        assert ndiff < 1e-3 * a.numel() and float(df.max()) <= 2 ** -7, (name, ndiff, float(df.max()))


@pytest.mark.parametrize('C,ln,T,H,W,bias', [(96, 1, 2, 24, 70, False), (48, 2, 1, 17, 41, True)])
def test_qkv3_hi_plane_only(ctx, C, ln, T, H, W, bias):
    """ira_set_weight_planes(1), the handles' default (r3): k_qkv3 reads the hi plane of the qkv 1x1 alone.  That is, to the bit, what the
    two-plane kernel computes on a blob whose lo plane is zero (x * 0 adds nothing) - the identity the accuracy ablation
    (tools/ablate_lo_restormer.py) and the handle's zeroed lo planes rest on."""
    from ira_amd.config import RestormerConfig
    from ira_amd.pack import pack_block
    g = torch.Generator().manual_seed(3 * C + ln + W)
    hid = 32
    sd = {
        'b.norm1.body.weight': 1 + 0.1 * torch.randn(C, generator=g), 'b.norm2.body.weight': torch.ones(C),
        'b.attn.temperature': torch.ones(1, 1, 1), 'b.attn.qkv.weight': torch.randn(3 * C, C, 1, 1, generator=g) / C ** 0.5,
        'b.attn.qkv_dwconv.weight': torch.randn(3 * C, 1, 3, 3, generator=g) / 3,
        'b.attn.project_out.weight': torch.zeros(C, C, 1, 1),
        'b.ffn.project_in.weight': torch.zeros(2 * hid, C, 1, 1), 'b.ffn.dwconv.weight': torch.zeros(2 * hid, 1, 3, 3),
        'b.ffn.project_out.weight': torch.zeros(C, hid, 1, 1),
    }
    if ln == 2:
        sd['b.norm1.body.bias'] = 0.1 * torch.randn(C, generator=g)
        sd['b.norm2.body.bias'] = torch.zeros(C)
    if bias:
        for k, n in [('attn.qkv', 3 * C), ('attn.qkv_dwconv', 3 * C), ('attn.project_out', C), ('ffn.project_in', 2 * hid),
                     ('ffn.dwconv', 2 * hid), ('ffn.project_out', C)]:
            sd[f'b.{k}.bias'] = 0.1 * torch.randn(n, generator=g)
    cfg = RestormerConfig(dim=C, LayerNorm_type='BiasFree' if ln == 1 else 'WithBias', bias=bias)
    cfg.hidden = lambda d_: hid
    arrs = {}
    pack_block(sd, 'b', C, hid, cfg, arrs, 'b')
    wz = arrs['b.qkv_w'].copy()
    assert wz.shape[0] == 2
    wz[1] = 0                                          # the blob with a zero lo plane
    d = {k: dev(torch.from_numpy(np.ascontiguousarray(v_))) for k, v_ in arrs.items()}
    wzd = dev(torch.from_numpy(np.ascontiguousarray(wz)))
    xd = (torch.randn(T, H, W, C, generator=g) * 1.5).cuda()
    WB = (W + 7) // 8

    def run(be, w):
        qk = torch.full((T, H, WB, 2 * C, 8), float('nan'), dtype=torch.float16).cuda()
        v = torch.full((T, H, W, C), float('nan'), dtype=torch.float16).cuda()
        rc = be._lib.ira_op_qkv_fused_gen(be.handle, 3, ptr(xd), ln, C, ptr(w), ptr(d.get('b.qkv_b')), ptr(d['b.dw1_w']), ptr(d.get('b.dw1_b')),
                                          ptr(qk), ptr(v), T, H, W)
        assert rc == 0, be._lib.ira_last_error(be.handle)
        torch.cuda.synchronize()
        return qk, v
    q2, v2 = run(ctx, wzd)                             # two-plane kernel, zero lo plane
    with _hi_plane_ctx() as c1:
        q1, v1 = run(c1, d['b.qkv_w'])                 # hi-plane kernel on the full blob
    assert torch.isfinite(q1.float()).all() and torch.isfinite(v1.float()).all()
    assert torch.equal(q1.view(torch.int16), q2.view(torch.int16)) and torch.equal(v1.view(torch.int16), v2.view(torch.int16))
