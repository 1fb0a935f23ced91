"""CPU ORACLE for the build-defined DnCNN (BASELINE config 1) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED versus the reference: /root/reference contains no DnCNN code, so there is nothing to
pin against.  This is the functional torch-CPU statement of the layout documented in ira_amd/dncnn.py; tests check it
against torch.nn modules (Conv2d / BatchNorm2d.eval() / ReLU) assembled from the same state dict."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dncnn_forward(x, sd, cfg, bn_eps=1e-5):
    nb = cfg['nb']
    h = F.relu(F.conv2d(x, sd['model.0.weight'], sd['model.0.bias'], padding=1))
    for k in range(nb - 2):
        c, b = 2 + 3 * k, 3 + 3 * k
        h = F.conv2d(h, sd[f'model.{c}.weight'], sd[f'model.{c}.bias'], padding=1)
        h = F.batch_norm(h, sd[f'model.{b}.running_mean'], sd[f'model.{b}.running_var'], sd[f'model.{b}.weight'], sd[f'model.{b}.bias'],
                         training=False, eps=bn_eps)
        h = F.relu(h)
    last = 2 + 3 * (nb - 2)
    return x - F.conv2d(h, sd[f'model.{last}.weight'], sd[f'model.{last}.bias'], padding=1)
