"""Times the fused SwinIR-L MLP half-block (ira_op_swin_mlp: x += fc2(GELU(fc1(LN(x)))), 256 padded channels, 480 hidden) on the token
count of the 544 x 968 bench frame, next to the two kernels it replaces (ira_op_pw_kres_ex + ira_op_pw_nres through an fp16 hidden tensor).
PLANES=1|2 picks the weight planes; IRA_SWIN_MLP_DBG=1..5 the measurement variants of the fused kernel (wrong results)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
import torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import reduced_config
from ira_amd.weights import seeded_state_dict
cfg0 = reduced_config(); be = RestormerBackend(cfg0, seeded_state_dict(cfg0, 1), weight_planes=int(os.environ.get("PLANES", "1"))); L = be._lib
p = lambda t: C.c_void_p(t.data_ptr())
npx = int(sys.argv[1]) if len(sys.argv) > 1 else 544 * 968
Cp, hid = 256, 480
x = torch.randn(npx, Cp, device='cuda'); x[:, 240:] = 0
W1 = (torch.randn(2, hid // 32, Cp // 16, 64, 8, device='cuda') * 0.05).half(); b1 = torch.randn(hid, device='cuda') * 0.1
W2 = (torch.randn(2, Cp // 32, hid // 16, 64, 8, device='cuda') * 0.05).half(); b2 = torch.randn(Cp, device='cuda') * 0.1
hbuf = torch.empty(npx, hid, device='cuda', dtype=torch.float16)
def timed(run, name, by):
    for _ in range(3): assert run() == 0, L.ira_last_error(be.handle)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * npx * Cp * hid * 2
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s (one plane)  {by / ms / 1e6:.0f} GB/s", flush=True)
fused = lambda: L.ira_op_swin_mlp(be.handle, p(x), p(W1), p(b1), p(W2), p(b2), Cp, hid, npx, C.c_float(1e-5), 240)
def two():
    rc = L.ira_op_pw_kres_ex(be.handle, p(x), 1, 2, Cp, p(W1), p(b1), p(hbuf), hid, hid, hid // 32, 1, npx, C.c_float(1e-5), 240, 1)
    return rc or L.ira_op_pw_nres(be.handle, p(hbuf), hid, None, 0, 0, p(W2), 0, p(b2), p(x), p(x), Cp, Cp // 32, 1, npx)
timed(fused, f"fused (dbg {os.environ.get('IRA_SWIN_MLP_DBG', '0')})", npx * Cp * 8)
att = torch.randn(npx, Cp, device='cuda').half()
Wp = (torch.randn(2, Cp // 32, Cp // 16, 64, 8, device='cuda') * 0.05).half(); bp = torch.randn(Cp, device='cuda') * 0.1
pfused = lambda: L.ira_op_swin_proj_mlp(be.handle, p(x), p(att), p(Wp), p(bp), p(W1), p(b1), p(W2), p(b2), Cp, hid, npx, C.c_float(1e-5), 240)
proj = lambda: L.ira_op_pw_nres(be.handle, p(att), Cp, None, 0, 0, p(Wp), 0, p(bp), p(x), p(x), Cp, Cp // 32, 1, npx)
if os.environ.get('IRA_SWIN_MLP_DBG', '0') == '0':
    timed(pfused, "proj + MLP fused", npx * Cp * 14)
    timed(proj, "proj + residual alone (k_nres_wide)", npx * Cp * 10)
if os.environ.get('IRA_SWIN_MLP_DBG', '0') == '0':
    timed(two, "two kernels", npx * (Cp * 12 + hid * 4))
