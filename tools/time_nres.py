"""Times the deep-level project_out + residual and attention-apply kernels (ira_op_pw_nres) at the headline shapes (40 tiles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, P, K, N, per_tile in (('L3 pout', 128 * 128, 512, 192, 0), ('L4 pout', 64 * 64, 1024, 384, 0), ('L3 apply', 128 * 128, 192, 192, 1), ('L4 apply', 64 * 64, 384, 384, 1)):
    NB = N // 32
    x = torch.randn(T * P, K, device='cuda').half()
    nW = T if per_tile else 1
    W = (torch.randn(nW, 2, NB, K // 16, 64, 8, device='cuda') * 0.05).half()
    r = torch.randn(T * P, N, device='cuda'); y = torch.empty_like(r)
    stride = 2 * NB * 32 * K if per_tile else 0
    run = lambda: L.ira_op_pw_nres(be.handle, p(x), K, None, 0, 0, p(W), stride, None, p(r), p(y), N, NB, T, P)
    for _ in range(3): assert run() == 0, L.ira_last_error(be.handle)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * T * P * K * N * 2
    by = T * P * (K * 2 + N * 8)
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s (hi+lo)  {by / ms / 1e6:.0f} GB/s")
