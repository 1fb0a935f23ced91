"""Timeline of one workgroup of k_pw_kres_lds (-DIRA_KRES_STAMPS build in tools/mb/libira_stamps.so) on the SwinIR-L qkv shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['IRA_HIP_LIB'] = os.path.join(ROOT, 'tools', 'mb', 'libira_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from ira_amd.backend import SwinIRBackend
from ira_amd.swinir import reduced_swinir_config, seeded_state_dict
from ira_amd.pack import frag_major, split16
cfg = reduced_swinir_config()
be = SwinIRBackend(cfg, seeded_state_dict(cfg, 1), device=0)
lib = be._lib
K, N, P = 256, 768, 526592
NB = N // 32
x = torch.randn(P, K).cuda()
W = torch.from_numpy(frag_major(split16(np.random.default_rng(0).standard_normal((NB * 32, K)) / K ** 0.5))).cuda()
y = torch.empty(P, N, dtype=torch.float16).cuda()
vp = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    rc = lib.ira_op_pw_kres_ex(be.handle, vp(x), 1, 2, K, vp(W), None, vp(y), N, N, NB, 1, P, 1e-5, 240, 0)
    assert rc == 0
n = 8 + 8 * NB
buf = (C.c_ulonglong * n)()
fn = C.CDLL(os.environ['IRA_HIP_LIB']).ira_debug_kres_stamps
fn.argtypes = [C.c_void_p, C.c_int]
print('rc', fn(buf, n))
t0 = buf[0]
print('prologue (LN, B fragments):', buf[1] - t0)
names = ['top', 'stg issued', 'mfma issued', 'stores issued', 'lds written', 'after barrier']
for nb in range(0, NB, 3):
    row = [buf[8 + 8 * nb + i] for i in range(6)]
    print(f"block {nb:2d}: start +{row[0] - t0:7d} | " + ' '.join(f"{names[i]}:{row[i] - row[i - 1]:5d}" for i in range(1, 6)))
print('total', buf[8 + 8 * (NB - 1) + 5] - t0, 'per block', (buf[8 + 8 * (NB - 1) + 5] - buf[8 + 5]) / (NB - 1))
