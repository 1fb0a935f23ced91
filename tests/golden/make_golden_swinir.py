"""Golden fixtures for SwinIR FROM THE REFERENCE MODULE (dev container only).  network_swinir.py:11 imports DropPath,
to_2tuple, trunc_normal_ from timm (not installed): a 3-symbol stub is injected, as SURVEY 8c describes (DropPath is the
identity in eval, trunc_normal_ only touches initial values that are overwritten by the seeded state dict)."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd'))
sys.path.insert(0, ROOT)
from ira_amd.swinir import SwinIRConfig, reduced_swinir_config, seeded_state_dict, task_config      # noqa: E402
from ira_amd.synth import synthetic_noisy_image                                                    # noqa: E402

REF = '/root/reference/AgenticIR/executor/super_resolution/tools/SwinIR/models/network_swinir.py'


def load_ref():
    layers = types.ModuleType('timm.models.layers')

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.):
            super().__init__()

        def forward(self, x):
            return x
    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    for name in ('timm', 'timm.models'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['timm.models.layers'] = layers
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_network_swinir', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.set_num_threads(8)
    mod = load_ref()
    spec = {}
    cases = {
        'reduced_dn_40x56': (reduced_swinir_config(), 1, 40, 56, 1),
        'reduced_dn_pad_37x50': (reduced_swinir_config(), 2, 37, 50, 2),
        'reduced_jpeg_w7_gray_35x49': (reduced_swinir_config(in_chans=1, window_size=7, img_size=28, img_range=255.), 3, 35, 49, 1),
        'reduced_realsr_x4_3conv_24x32': (reduced_swinir_config(upscale=4, upsampler='nearest+conv', resi_connection='3conv', embed_dim=64,
                                                                 num_heads=[2, 2]), 4, 24, 32, 1),
        'reduced_classical_x2_24x24': (reduced_swinir_config(upscale=2, upsampler='pixelshuffle'), 5, 24, 24, 1),
        'color_dn_48': (task_config('color_dn'), 6, 48, 48, 1),
        'real_sr_large_x4_32x40': (task_config('real_sr', 4, True), 7, 32, 40, 1),
    }
    for name, (cfg, seed, H, W, B) in cases.items():
        m = mod.SwinIR(**cfg.to_dict()).eval()
        sd = seeded_state_dict(cfg, seed)
        params = [k for k, _ in m.named_parameters()]
        assert params == list(sd.keys()), (name, [a for a in params if a not in sd][:5], [a for a in sd if a not in params][:5])
        spec[name] = [[k, list(v.shape)] for k, v in m.named_parameters()]
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.endswith('relative_position_index') or k.endswith('attn_mask') for k in missing), missing[:5]
        _, noisy = synthetic_noisy_image(H, W, cfg.in_chans, sigma=25.0, seed=400 + seed)
        x = torch.from_numpy(noisy).float().div(255.).permute(2, 0, 1).unsqueeze(0).contiguous()
        if B > 1:
            x = torch.cat([x, x.flip(-1)], 0)
        with torch.no_grad():
            y = m(x)
        np.savez_compressed(os.path.join(HERE, f'swinir_{name}.npz'), cfg=json.dumps(cfg.to_dict()), seed=seed,
                            input=x.numpy(), output=y.numpy().astype(np.float32))
        print(name, tuple(y.shape), float(y.abs().max()), float(y.std()), sum(p.numel() for p in m.parameters()))
    with open(os.path.join(HERE, 'swinir_state_dict_spec.json'), 'w') as f:
        json.dump({k: v for k, v in spec.items() if k in ('reduced_dn_40x56', 'reduced_realsr_x4_3conv_24x32', 'reduced_classical_x2_24x24')}, f)


if __name__ == '__main__':
    main()
