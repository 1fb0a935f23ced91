"""Golden fixtures for the pad-crop tiler FROM THE REFERENCE'S OWN TILER CODE (dev container only).

Two reference implementations of the same tiler are executed, not restated:

* ``XRestormerModel.pre_process / tile_process / test / post_process`` of
  AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/models/xrestormer_model.py:53-150.  The module imports pip ``basicsr`` (absent
  here) for its base class, registry and utilities, none of which these four methods touch: 1-symbol stubs are injected (as SURVEY 8c
  describes for the architecture files), the methods are called UNBOUND on a plain namespace that carries what they read (``opt``, ``lq``,
  ``net_g``), and the network is the imported reference ``XRestormer`` module with the build's seeded weights.
* ``RealESRGANer.pre_process / process / tile_process / post_process`` of
  AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/realesrganer.py:71-183 (imports cv2 and a download helper at module
  level, used by ``__init__`` / ``enhance`` only: stubbed), same treatment, with the imported reference ``RRDBNet`` as the network.

Stored: the float input image, the tiler options and the float output tensor of the reference methods.  The uint8 conversion around
the tensor path (tensor2img / cv2 in ``enhance``) is pip-basicsr / OpenCV code and stays outside the fixtures.
"""
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
sys.path.insert(0, HERE)
from ira_amd.synth import synthetic_noisy_image                                                   # noqa: E402

XR_MODEL = '/root/reference/AgenticIR/executor/dehazing/tools/X-Restormer/xrestormer/models/xrestormer_model.py'
ESRGANER = '/root/reference/AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/realesrganer.py'
RRDB = '/root/reference/AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/rrdbnet.py'


def _load(path, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_xrestormer_model_class():
    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls

    def stub(name, **attrs):
        m = sys.modules.setdefault(name, types.ModuleType(name))
        for k, v in attrs.items():
            setattr(m, k, v)
        return m
    nothing = lambda *a, **k: None      # noqa: E731
    stub('basicsr'); stub('basicsr.models', lr_scheduler=types.ModuleType('lr_scheduler'))
    stub('basicsr.models.lr_scheduler')
    stub('basicsr.utils', get_root_logger=nothing, imwrite=nothing, tensor2img=nothing)
    stub('basicsr.utils.registry', MODEL_REGISTRY=_Reg(), ARCH_REGISTRY=_Reg())
    stub('basicsr.models.sr_model', SRModel=type('SRModel', (), {}))
    stub('basicsr.archs', build_network=nothing)
    stub('basicsr.metrics', calculate_metric=nothing)
    stub('basicsr.losses', build_loss=nothing)
    stub('xrestormer'); stub('xrestormer.models', lr_scheduler=types.ModuleType('lr_scheduler'))
    stub('xrestormer.models.lr_scheduler')
    return _load(XR_MODEL, 'ref_xrestormer_model').XRestormerModel


def load_realesrganer_class():
    for name in ('cv2', 'utils', 'utils.file', 'utils.realesrgan', 'utils.realesrgan.rrdbnet'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['utils.file'].load_file_from_url = lambda *a, **k: None
    rr = _load(RRDB, 'ref_rrdbnet')
    sys.modules['utils.realesrgan.rrdbnet'].RRDBNet = rr.RRDBNet
    return _load(ESRGANER, 'ref_realesrganer').RealESRGANer, rr.RRDBNet


def run_xrestormer_tiler(cls, net, lq, padding_size, tile, scale):
    """What XRestormerModel.nondist_validation does per image between feed_data and get_current_visuals (:172-181): pre_process, then
    tile_process when the option file has a 'tile' section, else test(), then post_process."""
    opt = {'padding_size': padding_size, 'scale': scale}
    if tile is not None:
        opt['tile'] = {'tile_size': tile[0], 'tile_pad': tile[1]}
    obj = types.SimpleNamespace(opt=opt, lq=lq.clone(), net_g=net)
    cls.pre_process(obj)
    if 'tile' in opt:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):           # tile_process prints a line per tile
            cls.tile_process(obj)
    else:
        cls.test(obj)
    cls.post_process(obj)
    return obj.output


def run_realesrganer_tiler(cls, net, img_hwc, scale, tile, tile_pad, pre_pad):
    """RealESRGANer.enhance's tensor path (:206-212): pre_process, tile_process or process, post_process."""
    obj = types.SimpleNamespace(scale=scale, tile_size=tile, tile_pad=tile_pad, pre_pad=pre_pad, mod_scale=None, half=False, device='cpu', model=net)
    cls.pre_process(obj, img_hwc)
    with torch.no_grad():
        if tile > 0:
            cls.tile_process(obj)
        else:
            cls.process(obj)
    return cls.post_process(obj)


def main():
    torch.set_num_threads(8)
    # ---- X-Restormer harness
    from make_golden_xrestormer import load_ref as load_xr_arch
    from ira_amd.xrestormer import reduced_xrestormer_config, seeded_state_dict as xr_sd
    cls = load_xrestormer_model_class()
    arch = load_xr_arch()
    for name, cfg, seed, h, w, pad, tile in (
            ('xr_whole_odd_70x101', reduced_xrestormer_config(), 11, 70, 101, 64, None),               # reflect pad to 128 x 128, untiled
            ('xr_tiled_150x200', reduced_xrestormer_config(), 12, 150, 200, 64, (64, 64)),             # 3 x 4 cores with a 64-pixel halo each
            ('xr_tiled_sr_x4_40x52', reduced_xrestormer_config(scale=4), 13, 40, 52, 16, (16, 16)),    # SR preset: x4 outputs, padding 16
    ):
        net = arch.XRestormer(**cfg.to_dict()).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in xr_sd(cfg, seed).items()}, strict=True)
        _, noisy = synthetic_noisy_image(h, w, 3, sigma=25.0, seed=600 + seed)
        lq = torch.from_numpy(noisy).float().div(255.).permute(2, 0, 1).unsqueeze(0).contiguous()
        with torch.no_grad():
            out = run_xrestormer_tiler(cls, net, lq, pad, tile, cfg.scale)
        assert out.shape == (1, 3, h * cfg.scale, w * cfg.scale), out.shape
        np.savez_compressed(os.path.join(HERE, f'padcrop_{name}.npz'), kind='xrestormer', cfg=json.dumps(cfg.to_dict()), seed=seed, noisy=noisy,
                            padding_size=pad, tile_size=-1 if tile is None else tile[0], tile_pad=-1 if tile is None else tile[1],
                            output=out.numpy().astype(np.float32))
        print(name, tuple(out.shape), float(out.abs().mean()))
    # ---- RealESRGANer
    from ira_amd.rrdbnet import reduced_rrdbnet_config, seeded_state_dict as rr_sd
    ecls, RRDBNet = load_realesrganer_class()
    for name, cfg, seed, h, w, tile, tile_pad, pre_pad in (
            ('esrgan_x4_tiled_45x61', reduced_rrdbnet_config(scale=4), 21, 45, 61, 24, 6, 0),          # 2 x 3 tiles (pre_pad 0 as DiffBIR builds it)
            ('esrgan_x2_tiled_37x50', reduced_rrdbnet_config(scale=2), 22, 37, 50, 20, 4, 0),          # mod-pad 2 (odd sizes), pixel-unshuffle input
            ('esrgan_x1_whole_30x41', reduced_rrdbnet_config(scale=1), 23, 30, 41, 0, 10, 0),          # mod-pad 4, untiled
            ('esrgan_x2_prepad_33x47', reduced_rrdbnet_config(scale=2), 24, 33, 47, 16, 4, 5),         # pre_pad 5 + mod-pad: CPU oracle only
    ):
        net = RRDBNet(**cfg.to_dict()).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in rr_sd(cfg, seed).items()}, strict=True)
        _, noisy = synthetic_noisy_image(h, w, 3, sigma=10.0, seed=700 + seed)
        img = noisy.astype(np.float32) / 255.0
        out = run_realesrganer_tiler(ecls, net, img, cfg.scale, tile, tile_pad, pre_pad)
        assert out.shape == (1, 3, h * cfg.scale, w * cfg.scale), out.shape
        np.savez_compressed(os.path.join(HERE, f'padcrop_{name}.npz'), kind='realesrganer', cfg=json.dumps(cfg.to_dict()), seed=seed, noisy=noisy,
                            tile_size=tile, tile_pad=tile_pad, pre_pad=pre_pad, output=out.numpy().astype(np.float32))
        print(name, tuple(out.shape), float(out.abs().mean()))


if __name__ == '__main__':
    main()
