"""A/B of kernel variants that must be BITWISE equal (same operands, same MFMA order), selected by an environment switch:
    python tools/cmp_conv.py [IRA_CONV_TILE]     dense 3x3: pixel tile through LDS (1) vs per-tap global loads (0)
    python tools/cmp_conv.py IRA_PW_GEMM         1x1: both operands through LDS (1) vs per-wave pixel fragments (0)  Runs one MPRNet, one DnCNN and one SwinIR forward under each setting in a child process."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/image-restoration-agent_amd"); sys.path.insert(0, sys.argv[1])
from ira_amd.backend import MPRNetBackend, DnCNNBackend, SwinIRBackend
from ira_amd import mprnet, dncnn, swinir
g = np.random.default_rng(0)
out = {}
cfg = mprnet.reduced_mprnet_config(n_feat=40, scale_unetfeats=24, scale_orsnetfeats=24)
be = MPRNetBackend(cfg, mprnet.seeded_state_dict(cfg, 1)); out["mprnet"] = be.forward(g.random((2, 3, 40, 72), dtype=np.float32)); be.close()
cfg = dncnn.DnCNNConfig(); be = DnCNNBackend(cfg, dncnn.seeded_state_dict(cfg, 1)); out["dncnn"] = be.forward(g.random((1, cfg.in_nc, 45, 77), dtype=np.float32)); be.close()
from ira_amd.backend import NAFNetBackend
from ira_amd import nafnet
cfg = nafnet.reduced_nafnet_config(); be = NAFNetBackend(cfg, nafnet.seeded_state_dict(cfg, 1)); out["nafnet"] = be.forward(g.random((2, 3, 40, 56), dtype=np.float32)); be.close()
cfg = swinir.reduced_swinir_config(); be = SwinIRBackend(cfg, swinir.seeded_state_dict(cfg, 1)); out["swinir"] = be.forward(g.random((1, cfg.in_chans, 24, 40), dtype=np.float32)); be.close()
np.savez(sys.argv[2], **out)
'''


def main():
    var = sys.argv[1] if len(sys.argv) > 1 else 'IRA_CONV_TILE'
    outs = []
    for v in ('0', '1'):
        path = f'/tmp/cmp_conv_{v}.npz'
        env = dict(os.environ, **{var: v})
        subprocess.run([sys.executable, '-c', CHILD, ROOT, path], check=True, env=env)
        outs.append(np.load(path))
    ok = True
    for k in outs[0].files:
        same = np.array_equal(outs[0][k], outs[1][k])
        print(k, outs[0][k].shape, 'bitwise equal' if same else f'DIFFERENT max {np.abs(outs[0][k] - outs[1][k]).max()}')
        ok &= same
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
