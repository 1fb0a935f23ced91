"""A/B of the fused GDFN generations at the dominant shape (C = 96, Hf = 255, T x 512 x 512), interleaved in one process:
generation 1 = attention-apply kernel (k_pw_gemm) + ira_op_ffn_fused, generation 2 = ira_op_ffn2 with the apply folded in.
usage: python tools/time_ffn2.py [T=8] [C=96]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config, RestormerConfig
from ira_amd.weights import seeded_state_dict
from ira_amd.pack import pack_block, frag_major, split16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 96
hid = int(os.environ.get("FFN_HID", int(Cc * 2.66)))
H = W = 512
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2")))
Hp = (hid + 31) // 32 * 32
NBc = (Cc + 31) // 32
g = torch.Generator().manual_seed(0)
sd = {'b.norm1.body.weight': torch.ones(Cc), 'b.norm2.body.weight': torch.ones(Cc), 'b.attn.temperature': torch.ones(1, 1, 1),
      'b.attn.qkv.weight': torch.zeros(3 * Cc, Cc, 1, 1), 'b.attn.qkv_dwconv.weight': torch.zeros(3 * Cc, 1, 3, 3),
      'b.attn.project_out.weight': torch.zeros(Cc, Cc, 1, 1),
      'b.ffn.project_in.weight': torch.randn(2 * hid, Cc, 1, 1, generator=g) / Cc ** 0.5,
      'b.ffn.dwconv.weight': torch.randn(2 * hid, 1, 3, 3, generator=g) / 3,
      'b.ffn.project_out.weight': torch.randn(Cc, hid, 1, 1, generator=g) / hid ** 0.5}
cfg = RestormerConfig(dim=Cc, LayerNorm_type='BiasFree'); cfg.hidden = lambda d: hid
arrs = {}; pack_block(sd, 'b', Cc, hid, cfg, arrs, 'b')
d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in arrs.items()}
x = torch.randn(T, H, W, Cc, generator=g).cuda(); out = torch.empty_like(x); x1 = x.clone()
v = torch.randn(T, H, W, Cc, generator=g).half().cuda()
Mp = np.zeros((T, NBc * 32, Cc)); Mp[:, :Cc] = (torch.randn(T, Cc, Cc, generator=g) / Cc ** 0.5).numpy()
Md = torch.from_numpy(np.ascontiguousarray(np.transpose(frag_major(split16(Mp)), (1, 0, 2, 3, 4, 5)))).cuda()
p = lambda t: C.c_void_p(t.data_ptr())
stride = 2 * NBc * 32 * Cc
L = be._lib
def gen1():
    assert L.ira_op_pw_nres(be.handle, p(v), Cc, None, 0, 0, p(Md), stride, None, p(x), p(x1), Cc, NBc, T, H * W) == 0
    assert L.ira_op_ffn_fused(be.handle, p(x1), p(out), 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_w']), None, p(d['b.pout_w']), None, T, H, W, Hp) == 0
def gen1_ffn_only():
    assert L.ira_op_ffn_fused(be.handle, p(x1), p(out), 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_w']), None, p(d['b.pout_w']), None, T, H, W, Hp) == 0
def gen2():
    assert L.ira_op_ffn2(be.handle, p(x), p(out), p(v), p(Md), stride, None, 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_u']), None, p(d['b.pout_w']), None, T, H, W, Hp) == 0
def gen2_nofold():
    assert L.ira_op_ffn2(be.handle, p(x), p(out), None, None, 0, None, 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_u']), None, p(d['b.pout_w']), None, T, H, W, Hp) == 0
if os.environ.get('IRA_FFN2_DBG'):
    for _ in range(2): gen2_nofold()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gen2_nofold()
    torch.cuda.synchronize()
    print('IRA_FFN2_DBG=%s T=%d C=%d no fold: %.3f ms' % (os.environ['IRA_FFN2_DBG'], T, Cc, (time.perf_counter() - t0) * 100))
    if os.environ['IRA_FFN2_DBG'] in ('4', '9'):
        buf = (C.c_ulonglong * 128)()
        assert L.ira_debug_ffn2_stamps(buf) == 0
        P_ = [buf[i] for i in range(64)]; C_ = [buf[64 + i] for i in range(64)]
        t0 = min(P_[0], C_[0])
        print('P wave 0 stamps:', [int(v_ - t0) for v_ in P_[:32] if v_])
        print('C wave 4 stamps:', [int(v_ - t0) for v_ in C_[:32] if v_])
        print('P slot stamps (interval 5: before wait, after wait, ...):', [int(v_ - t0) for v_ in P_[32:] if v_])
        print('C slot stamps:', [int(v_ - t0) for v_ in C_[32:] if v_])
    sys.exit(0)
def gen2_lo(mask):
    def f():
        assert L.ira_op_ffn2_planes(be.handle, mask, p(x), p(out), p(v), p(Md), stride, None, 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_u']), None, p(d['b.pout_w']), None, T, H, W, Hp) == 0
    return f
fns = [('gen1 apply+ffn_fused', gen1), ('gen1 ffn_fused only', gen1_ffn_only), ('gen2 ffn2 with fold', gen2), ('gen2 ffn2 no fold', gen2_nofold)] + ([('gen2 fold, W_out plain fp16', gen2_lo(1)), ('gen2 fold, both plain fp16', gen2_lo(0))] if Cc == 96 else [])
for _, f in fns: f()
torch.cuda.synchronize()
o1 = None
gen1(); torch.cuda.synchronize(); o1 = out.clone(); gen2(); torch.cuda.synchronize()
print('max |gen1 - gen2| =', float((o1 - out).abs().max()), ' branch magnitude', float((o1 - x).abs().max()))
res = {n: [] for n, _ in fns}
for rnd in range(5):
    for n, f in fns:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); res[n].append((time.perf_counter() - t0) / 5 * 1e3)
for n, _ in fns:
    print(f'T={T} C={Cc} {n:28s}: median {sorted(res[n])[2]:.3f} ms  min {min(res[n]):.3f} ms')
