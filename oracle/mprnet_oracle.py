"""CPU ORACLE for MPRNet -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of AgenticIR/executor/denoising/tools/MPRNet/Deblurring/MPRNet.py (the Denoising / Deraining
copies differ only in the constructor defaults, :239): CALayer :21-37, CAB :42-57, SAM :61-74, Encoder :79-121, Decoder :123-151,
Down/Up/SkipUpSample :155-184, ORB :188-199, ORSNet :202-233, MPRNet.forward :263-347.  Its harness (demo.py:64-87) is the Restormer
harness without a tiler (pad to 8 by reflection only if needed, model(input_)[0], clamp, crop, img_as_ubyte): oracle/restormer_oracle.py
``restore_image(..., forward_fn=...)``.  Pinned against the reference module (tests/golden/make_golden_mprnet.py; plain torch, no stubs)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _c(x, w, k=3):
    return F.conv2d(x, w, None, padding=k // 2)


def cab(x, sd, p):
    """CAB.forward (:53-57) with CALayer (:34-37)."""
    r = _c(F.prelu(_c(x, sd[f'{p}.body.0.weight']), sd[f'{p}.body.1.weight']), sd[f'{p}.body.2.weight'])
    y = F.adaptive_avg_pool2d(r, 1)
    y = torch.sigmoid(F.conv2d(F.relu(F.conv2d(y, sd[f'{p}.CA.conv_du.0.weight'])), sd[f'{p}.CA.conv_du.2.weight']))
    return r * y + x


def _down(x, w):
    return F.conv2d(F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=False), w)       # DownSample :155-163


def _up(x, w):
    return F.conv2d(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), w)          # UpSample :165-173


def encoder(x, sd, p, enc_outs=None, dec_outs=None):
    """Encoder.forward (:104-121)."""
    outs = []
    for lvl in (1, 2, 3):
        for k in range(2):
            x = cab(x, sd, f'{p}.encoder_level{lvl}.{k}')
        if enc_outs is not None:
            x = x + F.conv2d(enc_outs[lvl - 1], sd[f'{p}.csff_enc{lvl}.weight']) + F.conv2d(dec_outs[lvl - 1], sd[f'{p}.csff_dec{lvl}.weight'])
        outs.append(x)
        if lvl < 3:
            x = _down(x, sd[f'{p}.down{lvl}{lvl + 1}.down.1.weight'])
    return outs


def decoder(outs, sd, p):
    """Decoder.forward (:141-151)."""
    enc1, enc2, enc3 = outs
    d3 = enc3
    for k in range(2):
        d3 = cab(d3, sd, f'{p}.decoder_level3.{k}')
    x = _up(d3, sd[f'{p}.up32.up.1.weight']) + cab(enc2, sd, f'{p}.skip_attn2')
    d2 = x
    for k in range(2):
        d2 = cab(d2, sd, f'{p}.decoder_level2.{k}')
    x = _up(d2, sd[f'{p}.up21.up.1.weight']) + cab(enc1, sd, f'{p}.skip_attn1')
    d1 = x
    for k in range(2):
        d1 = cab(d1, sd, f'{p}.decoder_level1.{k}')
    return [d1, d2, d3]


def sam(x, x_img, sd, p):
    """SAM.forward (:68-74), kernel_size 1 (:256-257)."""
    x1 = F.conv2d(x, sd[f'{p}.conv1.weight'])
    img = F.conv2d(x, sd[f'{p}.conv2.weight']) + x_img
    x2 = torch.sigmoid(F.conv2d(img, sd[f'{p}.conv3.weight']))
    return x1 * x2 + x, img


def orsnet(x, enc, dec, sd, cfg):
    """ORSNet.forward (:224-233) with ORB (:196-199)."""
    p = 'stage3_orsnet'

    def orb(x, i):
        r = x
        for k in range(cfg['num_cab']):
            r = cab(r, sd, f'{p}.orb{i}.body.{k}')
        return _c(r, sd[f'{p}.orb{i}.body.{cfg["num_cab"]}.weight']) + x
    x = orb(x, 1)
    x = x + F.conv2d(enc[0], sd[f'{p}.conv_enc1.weight']) + F.conv2d(dec[0], sd[f'{p}.conv_dec1.weight'])
    x = orb(x, 2)
    x = x + F.conv2d(_up(enc[1], sd[f'{p}.up_enc1.up.1.weight']), sd[f'{p}.conv_enc2.weight']) \
          + F.conv2d(_up(dec[1], sd[f'{p}.up_dec1.up.1.weight']), sd[f'{p}.conv_dec2.weight'])
    x = orb(x, 3)
    ue = _up(_up(enc[2], sd[f'{p}.up_enc2.0.up.1.weight']), sd[f'{p}.up_enc2.1.up.1.weight'])
    ud = _up(_up(dec[2], sd[f'{p}.up_dec2.0.up.1.weight']), sd[f'{p}.up_dec2.1.up.1.weight'])
    return x + F.conv2d(ue, sd[f'{p}.conv_enc3.weight']) + F.conv2d(ud, sd[f'{p}.conv_dec3.weight'])


def mprnet_forward(x3, sd, cfg, all_stages=False):
    """MPRNet.forward (:263-347).  Returns stage3_img + x3 (the tensor demo.py uses: restored[0]); H, W multiples of 8."""
    H, W = x3.shape[2:]
    top, bot = x3[:, :, :H // 2], x3[:, :, H // 2:]
    quads = [top[:, :, :, :W // 2], top[:, :, :, W // 2:], bot[:, :, :, :W // 2], bot[:, :, :, W // 2:]]
    shallow = lambda img, i: cab(_c(img, sd[f'shallow_feat{i}.0.weight']), sd, f'shallow_feat{i}.1')
    f1 = [encoder(shallow(q, 1), sd, 'stage1_encoder') for q in quads]
    f1_top = [torch.cat((a, b), 3) for a, b in zip(f1[0], f1[1])]
    f1_bot = [torch.cat((a, b), 3) for a, b in zip(f1[2], f1[3])]
    r1_top, r1_bot = decoder(f1_top, sd, 'stage1_decoder'), decoder(f1_bot, sd, 'stage1_decoder')
    s_top, img1_top = sam(r1_top[0], top, sd, 'sam12')
    s_bot, img1_bot = sam(r1_bot[0], bot, sd, 'sam12')
    x2_top = _c(torch.cat([shallow(top, 2), s_top], 1), sd['concat12.weight'])
    x2_bot = _c(torch.cat([shallow(bot, 2), s_bot], 1), sd['concat12.weight'])
    f2_top = encoder(x2_top, sd, 'stage2_encoder', f1_top, r1_top)
    f2_bot = encoder(x2_bot, sd, 'stage2_encoder', f1_bot, r1_bot)
    f2 = [torch.cat((a, b), 2) for a, b in zip(f2_top, f2_bot)]
    r2 = decoder(f2, sd, 'stage2_decoder')
    s3, img2 = sam(r2[0], x3, sd, 'sam23')
    x3c = _c(torch.cat([shallow(x3, 3), s3], 1), sd['concat23.weight'])
    x3c = orsnet(x3c, f2, r2, sd, cfg)
    out = _c(x3c, sd['tail.weight']) + x3
    if all_stages:
        return [out, img2, torch.cat([img1_top, img1_bot], 2)]
    return out
