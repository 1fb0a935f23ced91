"""Time ira_op_qkv_fused alone (C=96, 8 x 512x512); with the stamps build also print the s_memtime timeline."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config, RestormerConfig
from ira_amd.weights import seeded_state_dict
from ira_amd.pack import pack_block
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2")))
Cc, hid, T, H, W = int(os.environ.get('QC', '96')), 32, int(os.environ.get('QT', '8')), 512, 512
GEN = int(os.environ.get('GEN', '3'))
g = torch.Generator().manual_seed(0)
sd = {'b.norm1.body.weight': torch.ones(Cc), 'b.norm2.body.weight': torch.ones(Cc), 'b.attn.temperature': torch.ones(1, 1, 1),
      'b.attn.qkv.weight': torch.randn(3 * Cc, Cc, 1, 1, generator=g) / Cc ** 0.5, 'b.attn.qkv_dwconv.weight': torch.randn(3 * Cc, 1, 3, 3, generator=g) / 3,
      'b.attn.project_out.weight': torch.zeros(Cc, Cc, 1, 1), 'b.ffn.project_in.weight': torch.zeros(2 * hid, Cc, 1, 1),
      'b.ffn.dwconv.weight': torch.zeros(2 * hid, 1, 3, 3), 'b.ffn.project_out.weight': torch.zeros(Cc, hid, 1, 1)}
cfg = RestormerConfig(dim=Cc, LayerNorm_type='BiasFree'); cfg.hidden = lambda d: hid
arrs = {}; pack_block(sd, 'b', Cc, hid, cfg, arrs, 'b')
d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in arrs.items()}
x = torch.randn(T, H, W, Cc, generator=g).cuda()
qk = torch.empty(T, H, W // 8, 2 * Cc, 8, dtype=torch.float16, device='cuda'); v = torch.empty(T, H, W, Cc, dtype=torch.float16, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())
def run():
    rc = be._lib.ira_op_qkv_fused_gen(be.handle, GEN, p(x), 1, Cc, p(d['b.qkv_w']), None, p(d['b.dw1_w']), None, p(qk), p(v), T, H, W); assert rc == 0
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); print('qkv_fused gen %d C=%d T=%d dbg=%s: %.3f ms' % (GEN, Cc, T, os.environ.get('IRA_QKV3_DBG', os.environ.get('QDBG', '0')), (time.perf_counter() - t0) * 100))
if os.environ.get('IRA_HIP_LIB', '').endswith('dbg.so') and GEN == 3:
    buf = (C.c_ulonglong * 64)(); lib = C.CDLL(os.environ['IRA_HIP_LIB']); assert lib.ira_debug_qkv3_stamps(buf) == 0
    st = [int(buf[i] - buf[0]) for i in range(64)]
    print('wave 0 of workgroup 1777: start, pixels + weights requested, group 0 normalised, all normalised, then per interval (work done, barrier passed):')
    print('   ', st[:4], [tuple(st[4 + 2 * k: 6 + 2 * k]) for k in range(10)])