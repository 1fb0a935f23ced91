"""Time ira_op_ffn_fused alone at the dominant shape (C=96, 8 x 512x512).  IRA_FFN_DBG selects ablations."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config, RestormerConfig
from ira_amd.weights import seeded_state_dict
from ira_amd.pack import pack_block
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2")))
Cc, hid, T, H, W = int(os.environ.get('C', 96)), int(os.environ.get('HID', 255)), 8, 512, 512
Hp = (hid + 31) // 32 * 32
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
x = torch.randn(T, H, W, Cc, generator=g).cuda(); out = torch.empty_like(x)
p = lambda t: C.c_void_p(t.data_ptr())
def run():
    rc = be._lib.ira_op_ffn_fused(be.handle, p(x), p(out), 1, Cc, p(d['b.pin_w']), None, p(d['b.dw2_w']), None, p(d['b.pout_w']), None, T, H, W, Hp)
    assert rc == 0
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
print('IRA_FFN_DBG=%s C=%d: %.3f ms' % (os.environ.get('IRA_FFN_DBG', '0'), Cc, (time.perf_counter() - t0) * 100))
if os.environ.get('IRA_HIP_LIB', '').endswith('stamps.so'):
    buf = (C.c_ulonglong * 128)()
    lib = C.CDLL(os.environ['IRA_HIP_LIB'])
    assert lib.ira_debug_ffn_stamps(buf) == 0
    prod = [buf[i] for i in range(64)]; cons = [buf[64 + i] for i in range(64)]
    t0 = min(prod[0], cons[0])
    print('producer stamps (cycles@100MHz? raw delta):', [int(v - t0) for v in prod[:20]])
    print('consumer stamps:', [int(v - t0) for v in cons[:20]], 'end', int(cons[30] - t0))
