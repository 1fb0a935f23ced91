"""Accuracy side of dropping the lo weight plane per layer type in SwinIR (review item r1-8): the two full-preset goldens of
tests/golden/make_golden_swinir.py (colour denoise M at 48 x 48, real-SR L x4 at 32 x 40) through the HIP path with the lo plane of the
named weights zeroed (IRA_ABLATE_LO, a packer hook).  Prints PSNR(build, reference float) and |dPSNR| on uint8 for every subset.
usage (GPU box): python tools/ablate_lo_swinir.py"""
import glob, json, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'image-restoration-agent_amd')); sys.path.insert(0, R)
import numpy as np
from ira_amd.backend import SwinIRBackend
from ira_amd.swinir import SwinIRConfig, seeded_state_dict
from oracle import restormer_oracle as O

cases = [p for p in sorted(glob.glob(os.path.join(R, 'tests', 'golden', 'swinir_*.npz'))) if 'color_dn_48' in p or 'real_sr_large' in p]
subsets = ['', 'qkv_w', 'proj_w', 'fc1_w', 'fc2_w', 'fc1_w,fc2_w', 'proj_w,fc2_w', 'qkv_w,proj_w,fc1_w,fc2_w']
u8 = lambda x: O.img_as_ubyte(np.clip(x, 0, 1))
for path in cases:
    z = np.load(path)
    cfg = SwinIRConfig(**json.loads(str(z['cfg'])))
    x, ref = z['input'], z['output']
    xa = np.repeat(np.repeat(x, cfg.upscale, 2), cfg.upscale, 3) if cfg.upscale > 1 else x
    for sub in subsets:
        os.environ['IRA_ABLATE_LO'] = sub
        with SwinIRBackend(cfg, seeded_state_dict(cfg, int(z['seed'])), device=0) as be:
            y = be.forward(x)
        d = O.psnr_u8(u8(y), u8(xa)) - O.psnr_u8(u8(ref), u8(xa))
        print(f"{os.path.basename(path)[7:-4]:28s} lo dropped: {sub or '(none)':28s} PSNR(build, reference) {O.psnr_f(y, ref):6.2f} dB   dPSNR_u8 {d:+.2e}", flush=True)
os.environ['IRA_ABLATE_LO'] = ''
