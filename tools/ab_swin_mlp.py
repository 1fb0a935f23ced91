"""A/B of the fused SwinIR MLP kernel (csrc/ira_swin_mlp.hip) against the two kernels it replaces: one SwinIR-L x4 forward of a
seeded 272 x 272 frame (73984 tokens, above the fused kernel's 32768-token threshold) per weight-plane mode; prints the SHA-256 and the
mean of the fp32 output.  Run once with IRA_SWIN_MLP=0 and once with IRA_SWIN_MLP=1 (the switch is read once per process).  With the
erf-polynomial GELU in both paths the two runs were bit-identical (commit edbe53c); since the fused kernel takes GELU from the Phi table
the hashes differ and the means agree to the printed digits (tests/test_gpu_ops.py::test_swin_mlp_fused bounds the difference per element).
IRA_SWIN_PROJ_MLP=0/1 (projection as its own launch / folded into the fused kernel) must and does give the SAME hashes.
usage (GPU box): IRA_SWIN_MLP=0 python tools/ab_swin_mlp.py; IRA_SWIN_MLP=1 python tools/ab_swin_mlp.py; IRA_SWIN_PROJ_MLP=0 python tools/ab_swin_mlp.py"""
import hashlib, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'image-restoration-agent_amd')); sys.path.insert(0, R)
import numpy as np
from ira_amd.backend import SwinIRBackend
from ira_amd.swinir import seeded_state_dict, task_config

cfg = task_config('real_sr', 4, True)
x = np.random.default_rng(5).random((1, 3, 272, 272), dtype=np.float32)
for planes in (1, 2):
    with SwinIRBackend(cfg, seeded_state_dict(cfg, 0), device=0, weight_planes=planes) as be:
        y = be.forward(x)
    print(f"planes {planes}  shape {y.shape}  sha256 {hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()}  mean {float(y.mean()):.6f}", flush=True)
