"""Times the level-3/4 LayerNorm + 1x1 kernel (ira_op_pw_kres_ex) at the headline shapes.  usage: python tools/time_pw.py [act bits]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
act = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
for name, T, P, K, N, f32, ln in (('L3 ln_pin', 40, 128 * 128, 192, 1024, 1, 1), ('L4 ln_pin', 40, 64 * 64, 384, 2048, 1, 1),
                                  ('L3 ln_qkv', 40, 128 * 128, 192, 576, 1, 1), ('L4 ln_qkv', 40, 64 * 64, 384, 1152, 1, 1),
                                  ('L3 pin f32 noLN', 40, 128 * 128, 192, 1024, 1, 0), ('L3 pin f16 noLN', 40, 128 * 128, 192, 1024, 0, 0),
                                  ('L4 pin f32 noLN', 40, 64 * 64, 384, 2048, 1, 0), ('L4 pin f16 noLN', 40, 64 * 64, 384, 2048, 0, 0)):
    NB = N // 32
    x = torch.randn(T * P, K, device='cuda')
    if not f32: x = x.half()
    W = (torch.randn(2, NB, K // 16, 64, 8, device='cuda') * 0.05).half()
    y = torch.empty(T * P, N, device='cuda', dtype=torch.float16)
    run = lambda: L.ira_op_pw_kres_ex(be.handle, p(x), f32, ln, K, p(W), None, p(y), N, N, NB, T, P, C.c_float(1e-5), 0, act)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * T * P * K * N * (1 if act & 0x200 else 2)
    by = T * P * (K * 4 + N * 2)
    print(f"{name} act={act:#x}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s (hi+lo counted)  {by / ms / 1e6:.0f} GB/s")
