"""Generates tests/golden/rrdbnet_*.npz by running the REFERENCE module (imported from /root/reference, plain torch) on seeded weights.
Run in the dev container only:  python tests/golden/make_golden_rrdbnet.py"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd'))
from ira_amd.rrdbnet import RRDBNetConfig, realesrgan_config, reduced_rrdbnet_config, seeded_state_dict, state_dict_spec   # noqa: E402

REF = '/root/reference/AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/rrdbnet.py'
spec = importlib.util.spec_from_file_location('ref_rrdbnet', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CASES = [('reduced_x4_20x28', reduced_rrdbnet_config(), 1, (2, 3, 20, 28)),
         ('reduced_x2_24x36', reduced_rrdbnet_config(scale=2, num_feat=64), 2, (1, 3, 24, 36)),
         ('reduced_x1_32x24', reduced_rrdbnet_config(scale=1, num_block=1), 3, (1, 3, 32, 24)),
         ('realesrgan_x2plus_32', realesrgan_config(2), 4, (1, 3, 32, 32))]
specs = {}
for name, cfg, seed, shape in CASES:
    net = ref.RRDBNet(**cfg.to_dict()).eval()
    sd = seeded_state_dict(cfg, seed)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(s)) for k, s in state_dict_spec(cfg)]
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = np.random.default_rng(seed).random(shape, dtype=np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
    print(name, y.shape, 'params', sum(p.numel() for p in net.parameters()), 'out rms', float(np.sqrt((y ** 2).mean())))
    np.savez_compressed(os.path.join(HERE, f'rrdbnet_{name}.npz'), input=x, output=y, cfg=json.dumps(cfg.to_dict()), seed=seed)
    if name.startswith('reduced_x4'):
        specs[name] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
with open(os.path.join(HERE, 'rrdbnet_state_dict_spec.json'), 'w') as f:
    json.dump(specs, f)
