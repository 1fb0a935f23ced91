"""Timeline of one workgroup of k_pw_nres_lds (built with -DIRA_NRES_STAMPS into tools/mb/libira_stamps.so) on the SwinIR-L fc2 shape."""
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
K, N, NB, P = int(sys.argv[1]) if len(sys.argv) > 1 else 480, 256, 8, 526592
x = torch.randn(P, K).half().cuda()
W = torch.from_numpy(frag_major(split16(np.random.default_rng(0).standard_normal((NB * 32, K)) / K ** 0.5))).cuda()
res = torch.randn(P, N).cuda()
vp = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    rc = lib.ira_op_pw_nres(be.handle, vp(x), K, None, 0, 0, vp(W), 0, None, vp(res), vp(res), N, NB, 1, P)
    assert rc == 0
n = 8 + 8 * (K // 16)
buf = (C.c_ulonglong * n)()
fn = lib.ira_debug_nres_stamps if hasattr(lib, 'ira_debug_nres_stamps') else C.CDLL(os.environ['IRA_HIP_LIB']).ira_debug_nres_stamps
fn.argtypes = [C.c_void_p, C.c_int]
print('rc', fn(buf, n))
t0 = buf[0]
names = ['top', 'stg issued', 'mfma issued', 'b issued', 'vmcnt0', 'lds written', 'after barrier']
prev = t0
for s in range(min(K // 16, 12)):
    row = [buf[8 + 8 * s + i] for i in range(7)]
    print(f"k-step {s:2d}: start +{row[0] - t0:7d} | " + ' '.join(f"{names[i]}:{row[i] - row[i - 1]:5d}" for i in range(1, 7)))
print('total k-loop cycles', buf[8 + 8 * (K // 16 - 1) + 6] - t0, 'per k-step', (buf[8 + 8 * (K // 16 - 1) + 6] - buf[8 + 6]) / (K // 16 - 1))
