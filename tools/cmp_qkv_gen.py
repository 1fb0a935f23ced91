"""Generation 1 against generation 2 of the fused attention front end on one shape: where and by how much do q,k / v differ?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config, RestormerConfig
from ira_amd.weights import seeded_state_dict
from ira_amd.pack import pack_block
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1))
Cc, hid = int(os.environ.get('QC', '96')), 32
T, H, W = [int(v) for v in os.environ.get('SHAPE', '4,96,224').split(',')]
g = torch.Generator().manual_seed(0)
sd = {'b.norm1.body.weight': torch.ones(Cc), 'b.norm2.body.weight': torch.ones(Cc), 'b.attn.temperature': torch.ones(1, 1, 1),
      'b.attn.qkv.weight': torch.randn(3 * Cc, Cc, 1, 1, generator=g) / Cc ** 0.5, 'b.attn.qkv_dwconv.weight': torch.randn(3 * Cc, 1, 3, 3, generator=g) / 3,
      'b.attn.project_out.weight': torch.zeros(Cc, Cc, 1, 1), 'b.ffn.project_in.weight': torch.zeros(2 * hid, Cc, 1, 1),
      'b.ffn.dwconv.weight': torch.zeros(2 * hid, 1, 3, 3), 'b.ffn.project_out.weight': torch.zeros(Cc, hid, 1, 1)}
cfg = RestormerConfig(dim=Cc, LayerNorm_type='BiasFree'); cfg.hidden = lambda d: hid
arrs = {}; pack_block(sd, 'b', Cc, hid, cfg, arrs, 'b')
d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in arrs.items()}
x = (torch.randn(T, H, W, Cc, generator=g) * 1.5).cuda()
WB = (W + 7) // 8
p = lambda t: C.c_void_p(t.data_ptr())
outs = []
for gen in (1, 2, 2):
    qk = torch.full((T, H, WB, 2 * Cc, 8), float('nan'), dtype=torch.float16, device='cuda'); v = torch.full((T, H, W, Cc), float('nan'), dtype=torch.float16, device='cuda')
    rc = be._lib.ira_op_qkv_fused_gen(be.handle, gen, p(x), 1, Cc, p(d['b.qkv_w']), None, p(d['b.dw1_w']), None, p(qk), p(v), T, H, W); assert rc == 0
    torch.cuda.synchronize(); outs.append((qk.float().cpu(), v.float().cpu()))
for name, k in (('qk', 0), ('v', 1)):
    a, b, c2 = outs[0][k], outs[1][k], outs[2][k]
    df = (a - b).abs()
    idx = (df > 0).nonzero()
    print(name, 'differ', len(idx), 'of', a.numel(), 'max |d|', float(df.max()), 'max |a|', float(a.abs().max()), ' gen2 run-to-run differing:', int((b != c2).sum()))
    if len(idx):
        rel = df[df > 0] / a.abs()[df > 0].clamp_min(1e-6)
        print('   relative diffs: median %.2e max %.2e' % (float(rel.median()), float(rel.max())))
        print('   first:', idx[:12].tolist())
        for dim in range(idx.shape[1]):
            vals, cnts = torch.unique(idx[:, dim], return_counts=True)
            print('   dim', dim, 'distinct', len(vals), 'top', sorted(zip(cnts.tolist(), vals.tolist()), reverse=True)[:8])
