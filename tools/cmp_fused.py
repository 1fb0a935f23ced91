"""Fused vs per-op Restormer paths against the fp32 oracle and the numerics-contract emulator (dev aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from ira_amd.backend import RestormerBackend
from ira_amd.config import task_config
from ira_amd.weights import seeded_state_dict, to_torch
from oracle import restormer_oracle as O
cfg = task_config('Real_Denoising')
sdn = seeded_state_dict(cfg, 7)
be = RestormerBackend(cfg, sdn, device=0)
g = torch.Generator().manual_seed(3)
x = torch.rand(2, 3, 64, 96, generator=g)
y1 = be.forward(x.numpy())
be.set_fused_ffn(False)
y0 = be.forward(x.numpy())
os.environ['X'] = '1'
with torch.no_grad():
    yo = O.restormer_forward(x, to_torch(sdn), cfg.to_dict()).numpy()
print('fused vs oracle', O.psnr_f(y1, yo), 'per-op vs oracle', O.psnr_f(y0, yo), 'fused vs per-op', O.psnr_f(y1, y0))
