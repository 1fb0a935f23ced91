"""Timeline of one workgroup (block 777, wave 0) of the fused SwinIR MLP kernel, from the -DIRA_MLP_STAMPS build in tools/mb/libira_stamps.so:
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DIRA_MLP_STAMPS -Iinclude -c image-restoration-agent_amd/csrc/ira_swin_mlp.hip -o /tmp/m.o
  and the link line of csrc/Makefile with /tmp/m.o in place of ira_swin_mlp.o.  s_memtime ticks."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['IRA_HIP_LIB'] = os.path.join(ROOT, 'tools', 'mb', 'libira_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd')); sys.path.insert(0, ROOT)
import torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "1"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
npx, Cp, hid = 544 * 968, 256, 480
x = torch.randn(npx, Cp, device='cuda'); x[:, 240:] = 0
W1 = (torch.randn(2, hid // 32, Cp // 16, 64, 8, device='cuda') * 0.05).half(); b1 = torch.randn(hid, device='cuda') * 0.1
W2 = (torch.randn(2, Cp // 32, hid // 16, 64, 8, device='cuda') * 0.05).half(); b2 = torch.randn(Cp, device='cuda') * 0.1
for _ in range(3):
    assert L.ira_op_swin_mlp(be.handle, p(x), p(W1), p(b1), p(W2), p(b2), Cp, hid, npx, C.c_float(1e-5), 240) == 0
torch.cuda.synchronize()
buf = (C.c_ulonglong * 160)()
fn = C.CDLL(os.environ['IRA_HIP_LIB']).ira_debug_mlp_stamps
fn.argtypes = [C.c_void_p, C.c_int]
print('rc', fn(buf, 160))
t0 = buf[0]
print(f"LN + B fragments {buf[1] - t0}; to first barrier {buf[2] - buf[1]}; loop {buf[3] - buf[2]}; epilogue {buf[4] - buf[3]}; total {buf[4] - t0}")
print(f"s_memrealtime (100 MHz) over the workgroup: {buf[6] - buf[5]} ticks = {(buf[6] - buf[5]) / 100:.1f} us -> s_memtime runs at {(buf[4] - t0) / ((buf[6] - buf[5]) / 100):.0f} MHz")
names = ['dma issued', 'fc1+gelu', 'fc2 issued', 'waitcnt', 'barrier']
for c in range(0, hid // 32 + 1):
    row = [buf[8 + 8 * c + i] for i in range(6)]
    print(f"trip {c:2d}: start +{row[0] - t0:7d} | " + ' '.join(f"{names[i - 1]}:{row[i] - row[i - 1]:5d}" for i in range(1, 6)) + f" | trip {row[5] - row[0]}")
