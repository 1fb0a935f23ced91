"""RRDBNet / Real-ESRGAN: the oracle restatement against outputs of the reference module
(tests/golden/rrdbnet_*.npz from tests/golden/make_golden_rrdbnet.py) and the state-dict layout against the reference's."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from ira_amd.rrdbnet import RRDBNetConfig, realesrgan_config, reduced_rrdbnet_config, seeded_state_dict, state_dict_spec
from oracle.rrdbnet_oracle import rrdbnet_forward

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(glob.glob(os.path.join(GOLD, 'rrdbnet_*.npz')))


def test_state_dict_spec_matches_reference():
    with open(os.path.join(GOLD, 'rrdbnet_state_dict_spec.json')) as f:
        ref = json.load(f)
    assert [[k, list(s)] for k, s in state_dict_spec(reduced_rrdbnet_config())] == ref['reduced_x4_20x28']
    n = lambda cfg: sum(int(np.prod(s)) for _, s in state_dict_spec(cfg))
    assert n(realesrgan_config(2)) == 16703171 and n(realesrgan_config(4)) == 16697987      # RealESRGAN_x2plus / x4plus
    with pytest.raises(ValueError):
        realesrgan_config(3)


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_oracle_matches_reference_module(path):
    g = np.load(path)
    cfgd = json.loads(str(g['cfg']))
    sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(RRDBNetConfig(**cfgd), int(g['seed'])).items()}
    with torch.no_grad():
        y = rrdbnet_forward(torch.from_numpy(g['input']), sd, cfgd).numpy()
    assert y.shape == g['output'].shape
    assert np.abs(y - g['output']).max() < 1e-4 * max(1.0, np.abs(g['output']).max())
