"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF (dev container only).

Run:  python tests/golden/make_golden.py
Needs /root/reference (read-only) -- it imports the reference ``Restormer`` module exactly as the
reference does (demo.py:95-96, ``runpy.run_path`` on restormer_arch.py), loads the build's seeded
weights into it with ``load_state_dict`` (demo.py:101-102 minus the file read) and records
inputs/outputs as small .npy/.json fixtures.  The reference never travels to the GPU box: only the
fixtures written here do.  No reference source text is stored, only tensors and name/shape lists.

Fixtures:
  restormer_state_dict_spec.json     names + shapes of the reference state_dict (3 task configs)
  net_<case>.npz                     network forward: seed, cfg, input (fp32 NCHW), output (fp32 NCHW)
  harness_<case>.npz                 harness (u8 HWC -> u8 HWC) with the reference network inside the
                                     restated harness (oracle.restore_image(forward_fn=ref_module))
"""
import json
import os
import sys
from runpy import run_path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd'))
sys.path.insert(0, ROOT)

from ira_amd.config import RestormerConfig, reduced_config, task_config   # noqa: E402
from ira_amd.weights import seeded_state_dict, to_torch                      # noqa: E402
from oracle import restormer_oracle as O                                     # noqa: E402

REF_ARCH = '/root/reference/AgenticIR/executor/defocus_deblurring/tools/Restormer/basicsr/models/archs/restormer_arch.py'


def ref_model(cfg: RestormerConfig, seed: int):
    arch = run_path(REF_ARCH)
    m = arch['Restormer'](**cfg.to_dict())
    m.load_state_dict(to_torch(seeded_state_dict(cfg, seed)), strict=True)
    return m.eval()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    arch = run_path(REF_ARCH)

    # ---- state-dict layout
    spec = {}
    for task in ('Real_Denoising', 'Motion_Deblurring', 'Gaussian_Gray_Denoising'):
        cfg = task_config(task)
        m = arch['Restormer'](**cfg.to_dict())
        spec[task] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    spec['reduced_BiasFree'] = [[k, list(v.shape)] for k, v in
                                arch['Restormer'](**reduced_config('BiasFree').to_dict()).state_dict().items()]
    with open(os.path.join(HERE, 'restormer_state_dict_spec.json'), 'w') as f:
        json.dump(spec, f)

    # ---- network forward cases
    net_cases = {
        # name: (cfg, seed, H, W)
        'reduced_biasfree_64': (reduced_config('BiasFree'), 1, 64, 64),
        'reduced_withbias_48x80': (reduced_config('WithBias'), 2, 48, 80),
        'full_biasfree_64': (task_config('Real_Denoising'), 3, 64, 64),
        'full_withbias_64': (task_config('Motion_Deblurring'), 4, 64, 64),
        'full_gray_64': (task_config('Gaussian_Gray_Denoising'), 5, 64, 64),
        'full_biasfree_128': (task_config('Real_Denoising'), 0, 128, 128),
    }
    for name, (cfg, seed, H, W) in net_cases.items():
        _, noisy = O.synthetic_noisy_image(H, W, cfg.inp_channels, sigma=25.0, seed=100 + seed)
        x = torch.from_numpy(noisy).float().div(255.).permute(2, 0, 1).unsqueeze(0).contiguous()
        m = ref_model(cfg, seed)
        with torch.no_grad():
            y = m(x)
        np.savez_compressed(os.path.join(HERE, f'net_{name}.npz'),
                            cfg=json.dumps(cfg.to_dict()), seed=seed,
                            input=x.numpy(), output=y.numpy().astype(np.float32))
        print(name, tuple(y.shape), float(y.abs().max()), float((y - x).pow(2).mean().sqrt()))

    # ---- harness cases (reference network inside the restated harness)
    harness_cases = {
        # name: (cfg, seed, h, w, tile, overlap)
        'reduced_whole_mult8': (reduced_config('BiasFree'), 11, 64, 88, None, 32),
        'reduced_whole_oddpad': (reduced_config('WithBias'), 12, 75, 91, None, 32),
        'reduced_tiled_64_16': (reduced_config('BiasFree'), 13, 150, 203, 64, 16),
        'reduced_tiled_bigtile': (reduced_config('BiasFree'), 14, 70, 100, 512, 32),   # tile=min(tile,h,w)
        'reduced_gray_tiled': (reduced_config('BiasFree', inp_channels=1), 15, 97, 130, 48, 8),
        'full_tiled_64_32': (task_config('Real_Denoising'), 16, 96, 112, 64, 32),
    }
    for name, (cfg, seed, h, w, tile, ov) in harness_cases.items():
        clean, noisy = O.synthetic_noisy_image(h, w, cfg.inp_channels, sigma=25.0, seed=200 + seed)
        m = ref_model(cfg, seed)
        out = O.restore_image(noisy, None, None, tile=tile, tile_overlap=ov, forward_fn=lambda t: m(t))
        np.savez_compressed(os.path.join(HERE, f'harness_{name}.npz'),
                            cfg=json.dumps(cfg.to_dict()), seed=seed, tile=-1 if tile is None else tile,
                            overlap=ov, clean=clean, noisy=noisy, restored=out)
        print(name, out.shape, O.psnr_u8(out, clean), O.psnr_u8(noisy, clean))


if __name__ == '__main__':
    main()
