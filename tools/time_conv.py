"""Times the Restormer up / down 3x3 convolutions (ira_op_conv3x3, modes 2 / 1) at the headline shapes (40 tiles of 512 x 512)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "2"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, mode, H, K, N in (('up 2->1', 2, 256, 96, 192), ('up 3->2', 2, 128, 192, 384), ('up 4->3', 2, 64, 384, 768),
                            ('down 1->2', 1, 512, 48, 24), ('down 2->3', 1, 256, 96, 48), ('down 3->4', 1, 128, 192, 96)):
    NB = (N + 31) // 32
    x = torch.randn(T, H, H, K, device='cuda')
    W = (torch.randn(2, 9, NB * 32, K, device='cuda') * 0.02).half()
    out = torch.empty(T * H * H * N, device='cuda')
    run = lambda: L.ira_op_conv3x3(be.handle, p(x), p(W), None, None, p(out), mode, T, H, H, K, N, NB)
    for _ in range(2): assert run() == 0, L.ira_last_error(be.handle)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * T * H * H * 9 * K * N * 3
    by = T * H * H * (K + N) * 4
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s (3 MFMAs per product)  {by / ms / 1e6:.0f} GB/s")
