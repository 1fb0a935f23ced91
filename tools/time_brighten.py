"""Times ira_brighten_u8_device on a 3840 x 2160 frame resident in HBM (HIP events on torch's current stream, which the call is given)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'image-restoration-agent_amd'))
from ira_amd._capi import BRIGHTEN_TOOLS, check, load_library   # noqa: E402

lib = load_library()
h, w = 2160, 3840
x = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device='cuda')
y = torch.empty_like(x)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for tool, code in BRIGHTEN_TOOLS.items():
    for _ in range(3):
        check(lib.ira_brighten_u8_device(0, st, code, x.data_ptr(), h, w, y.data_ptr()), None, tool)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        check(lib.ira_brighten_u8_device(0, st, code, x.data_ptr(), h, w, y.data_ptr()), None, tool)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = h * w * (9 if code == 2 else 6)
    print(f"{tool}: {ms * 1e3:.1f} us / 4K frame, {byts / ms / 1e6:.0f} GB/s algorithmic ({byts / 1e6:.1f} MB)")
