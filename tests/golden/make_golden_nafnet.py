"""Golden fixtures for NAFNet FROM THE REFERENCE MODULE (dev container only; see make_golden.py).

The reference file imports ``basicsr.utils.registry.ARCH_REGISTRY`` (pip basicsr, not installed, used only as a
class decorator): a 1-symbol stub module is injected into sys.modules, as SURVEY 8c describes."""
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
from ira_amd.nafnet import NAFNetConfig, reduced_nafnet_config, seeded_state_dict      # noqa: E402
from ira_amd.synth import synthetic_noisy_image                                        # noqa: E402

REF = '/root/reference/AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/archs/NAFNet_arch.py'


def load_ref():
    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls
    for name in ('basicsr', 'basicsr.utils', 'basicsr.utils.registry'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['basicsr.utils.registry'].ARCH_REGISTRY = _Reg()
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_nafnet_arch', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.set_num_threads(8)
    mod = load_ref()
    spec = {}
    cases = {'reduced_40x56': (reduced_nafnet_config(), 1, 40, 56), 'reduced_pad_37x50': (reduced_nafnet_config(), 2, 37, 50),
             'w32_64': (NAFNetConfig(), 3, 64, 64), 'w32_96x128': (NAFNetConfig(), 4, 96, 128)}
    for name, (cfg, seed, H, W) in cases.items():
        m = mod.NAFNet(**cfg.to_dict()).eval()
        spec[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        sd = seeded_state_dict(cfg, seed)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        _, noisy = synthetic_noisy_image(H, W, 3, sigma=25.0, seed=300 + seed)
        x = torch.from_numpy(noisy).float().div(255.).permute(2, 0, 1).unsqueeze(0).contiguous()
        x = torch.cat([x, x.flip(-1)], 0)            # batch of 2: SCA statistics are per image
        with torch.no_grad():
            y = m(x)
        np.savez_compressed(os.path.join(HERE, f'nafnet_{name}.npz'), cfg=json.dumps(cfg.to_dict()), seed=seed,
                            input=x.numpy(), output=y.numpy().astype(np.float32))
        print(name, tuple(y.shape), float(y.abs().max()), float((y - x).pow(2).mean().sqrt()), sum(p.numel() for p in m.parameters()))
    with open(os.path.join(HERE, 'nafnet_state_dict_spec.json'), 'w') as f:
        json.dump({k: v for k, v in spec.items() if k in ('reduced_40x56', 'w32_64')}, f)


if __name__ == '__main__':
    main()
