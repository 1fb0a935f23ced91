"""A/B of the two dense 3x3 kernels (IRA_CONV_TILE=1: pixel tile through LDS; 0: per-tap global loads): the outputs must be BITWISE equal
(same operands, same MFMA order).  Runs one MPRNet, one DnCNN and one SwinIR forward under each setting in a child process."""
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
cfg = swinir.reduced_swinir_config(); be = SwinIRBackend(cfg, swinir.seeded_state_dict(cfg, 1)); out["swinir"] = be.forward(g.random((1, cfg.in_chans, 24, 40), dtype=np.float32)); be.close()
np.savez(sys.argv[2], **out)
'''


def main():
    outs = []
    for v in ('0', '1'):
        path = f'/tmp/cmp_conv_{v}.npz'
        env = dict(os.environ, IRA_CONV_TILE=v)
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
