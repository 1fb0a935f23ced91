"""Times the deep-level depthwise 3x3 kernels (ira_op_dwconv_qkv / ira_op_dwconv_gate) at the headline shapes (40 tiles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def timeit(run):
    for _ in range(2): assert run() == 0, L.ira_last_error(be.handle)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for name, H, Cc in (('L3', 128, 192), ('L4', 64, 384)):
    x = torch.randn(T, H, H, 3 * Cc, device='cuda').half()
    w = torch.randn(9, 3 * Cc, device='cuda')
    WB = (H + 7) // 8
    qk = torch.empty(T * H * WB * 2 * Cc * 8, device='cuda', dtype=torch.float16)
    v = torch.empty(T * H * H * Cc, device='cuda', dtype=torch.float16)
    ms = timeit(lambda: L.ira_op_dwconv_qkv(be.handle, p(x), p(w), None, p(qk), p(v), T, H, H, Cc))
    by = T * H * H * 3 * Cc * 2 * 2
    print(f"{name} dwconv_qkv: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s")
    Hp = (int(Cc * 2.66) + 31) // 32 * 32
    x2 = torch.randn(T, H, H, 2 * Hp, device='cuda').half()
    w2 = torch.randn(9, 2 * Hp, device='cuda')
    o2 = torch.empty(T * H * H * Hp, device='cuda', dtype=torch.float16)
    ms = timeit(lambda: L.ira_op_dwconv_gate(be.handle, p(x2), p(w2), None, p(o2), T, H, H, Hp))
    by = T * H * H * 3 * Hp * 2
    print(f"{name} dwconv_gate: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s")
