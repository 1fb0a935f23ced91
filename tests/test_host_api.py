"""Host-side mirror of the reference's tool / registry / service API (no GPU: a fake backend stands in
for RestormerBackend).  Reference behaviour: AgenticIR/executor/tool.py:33-62, executor/__init__.py:24-94,
multitask_tools.py:136-173, AgentApp/model_service/denoising/Restormer/model_serving.py:55-209."""
import io
import os
import re

import numpy as np
import pytest
from PIL import Image

from ira_amd import tools as T
from ira_amd import service as S
from ira_amd.config import SUPPORTED_TASKS, task_config


class FakeBackend:
    calls = []

    def __init__(self, cfg, sd, device):
        self.cfg = cfg

    def restore(self, img, tile=None, tile_overlap=32):
        FakeBackend.calls.append((img.shape, tile, tile_overlap))
        return (255 - img).astype(np.uint8)

    def close(self):
        pass


@pytest.fixture()
def cache():
    FakeBackend.calls = []
    return T.BackendCache(device=0, state_dict_loader=lambda task, wd=None: {}, backend_factory=FakeBackend)


def _img(h=21, w=30, c=3, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


def test_tool_contract(tmp_path, cache):
    inp, out = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir(); out.mkdir()
    img = _img()
    Image.fromarray(img).save(inp / 'input.png')
    tool = T.Restormer(subtask='denoising', cache=cache)
    assert tool.tool_name == 'restormer' and tool.subtask == 'denoising' and tool.opt_task == 'Real_Denoising'
    tool(inp, out, silent=True)
    assert os.listdir(out) == ['output.png']                      # tool.py:56-62
    got = np.asarray(Image.open(out / 'output.png'))
    assert np.array_equal(got, 255 - img)
    assert FakeBackend.calls == [((21, 30, 3), None, 32)]         # whole-image mode: no --tile (multitask_tools.py:159-165)
    # output dir must be empty, input dir must hold exactly one file (tool.py:51-54)
    with pytest.raises(AssertionError):
        tool(inp, out, silent=True)
    (inp / 'second.png').write_bytes((inp / 'input.png').read_bytes())
    out2 = tmp_path / 'out2'; out2.mkdir()
    with pytest.raises(AssertionError):
        tool(inp, out2, silent=True)


@pytest.mark.parametrize('subtask,task', [('denoising', 'Real_Denoising'), ('motion_deblurring', 'Motion_Deblurring'),
                                          ('defocus_deblurring', 'Single_Image_Defocus_Deblurring'), ('deraining', 'Deraining'),
                                          ('motion deblurring', 'Motion_Deblurring')])
def test_subtask_to_task(subtask, task, cache):
    assert T.Restormer(subtask=subtask, cache=cache).opt_task == task      # multitask_tools.py:151-156


def test_executor_registry_and_directory_protocol(tmp_path, cache):
    ex = T.build_executor(cache)
    assert ex.subtasks == {'denoising', 'motion deblurring', 'defocus deblurring', 'deraining', 'super-resolution',
                           'jpeg compression artifact removal', 'dehazing', 'brightening'}
    assert [t.tool_name for t in ex.toolbox_router['brightening']] == ['histogram_equalization', 'gamma_correction', 'constant_shift']
    # reference order, minus the tools this backend does not build (denoising/__init__.py:12-19, motion_deblurring :10-15, deraining :10-15)
    assert [t.tool_name for t in ex.toolbox_router['denoising']] == ['xrestormer', 'swinir_15', 'swinir_50', 'mprnet', 'restormer']
    assert [t.tool_name for t in ex.toolbox_router['motion deblurring']] == ['restormer', 'mprnet', 'xrestormer']
    assert [t.tool_name for t in ex.toolbox_router['deraining']] == ['xrestormer', 'restormer', 'mprnet']
    assert [t.tool_name for t in ex.toolbox_router['dehazing']] == ['xrestormer']
    assert [t.tool_name for t in ex.toolbox_router['super-resolution']] == ['xrestormer', 'swinir_gan', 'swinir_psnr']
    assert [t.tool_name for t in ex.toolbox_router['jpeg compression artifact removal']] == ['swinir_40']
    root = tmp_path / 'task'
    (root / '0-img').mkdir(parents=True)
    img = _img(seed=3)
    Image.fromarray(img).save(root / '0-img' / 'input.png')
    outp = ex.execute_subtask('defocus deblurring', root / '0-img' / 'input.png')     # executor/__init__.py:40-85
    assert outp == root / 'subtask1-defocus_deblurring' / 'tool1-restormer' / '0-img' / 'output.png'
    assert ex.executed_subtask_cnt == 1
    o2 = tmp_path / 'o2'; o2.mkdir()
    ex.invoke_a_tool('denoising', 'restormer', root / '0-img', o2)
    assert (o2 / 'output.png').exists()


def test_install_into_reference_executor(cache):
    class RefTool:
        def __init__(self, name, subtask):
            self.tool_name, self.subtask = name, subtask

    class RefExecutor:
        toolbox_router = {'denoising': [RefTool('xrestormer', 'denoising'), RefTool('restormer', 'denoising')],
                          'dehazing': [RefTool('maxim', 'dehazing')], 'deraining': [RefTool('maxim', 'deraining'), RefTool('mprnet', 'deraining')]}
        toolbox_router['super-resolution'] = [RefTool('diffbir', 'super_resolution'), RefTool('swinir_gan', 'super_resolution'),
                                              RefTool('xrestormer', 'super_resolution')]
    n = T.install_into(RefExecutor, cache)
    assert n == 5          # restormer, swinir_gan, xrestormer (denoising), xrestormer (super-resolution, scale 4), mprnet (deraining)
    assert isinstance(RefExecutor.toolbox_router['deraining'][1], T.MPRNet) and RefExecutor.toolbox_router['deraining'][1].opt_task == 'Deraining'
    assert RefExecutor.toolbox_router['deraining'][0].tool_name == 'maxim'
    assert isinstance(RefExecutor.toolbox_router['denoising'][0], T.XRestormer)
    assert RefExecutor.toolbox_router['super-resolution'][2].scale == 4
    tb = RefExecutor.toolbox_router['denoising']
    assert tb[0].tool_name == 'xrestormer' and isinstance(tb[1], T.Restormer) and tb[1].opt_task == 'Real_Denoising'
    sr = RefExecutor.toolbox_router['super-resolution']
    assert sr[0].tool_name == 'diffbir' and isinstance(sr[1], T.SwinIR) and sr[1].tool_name == 'swinir_gan' and sr[1].opt_task == 'real_sr'


def test_swinir_tool_contract(tmp_path):
    """multitask_tools.py:81-133: names, task / checkpoint mapping, x4 large model for SR, directory protocol."""
    made = []

    class FakeSR:
        def __init__(self, cfg, sd, device):
            self.cfg = cfg
            made.append((cfg.upscale, cfg.embed_dim, cfg.upsampler, cfg.resi_connection, cfg.window_size, sd))

        def restore(self, img, tile=None, tile_overlap=32):
            s = self.cfg.upscale
            return np.repeat(np.repeat(255 - img, s, 0), s, 1)
    sc = T.SwinIRCache(state_dict_loader=lambda task, name, wd: (task, name), backend_factory=FakeSR)
    tool = T.SwinIR('super_resolution', 'gan', cache=sc)
    assert (tool.tool_name, tool.opt_task, tool.model_name) == ('swinir_gan', 'real_sr', '003_realSR_BSRGAN_DFOWMFC_s64w8_SwinIR-L_x4_GAN.pth')
    inp, out = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir(); out.mkdir()
    img = _img(10, 12)
    Image.fromarray(img).save(inp / 'input.png')
    tool(inp, out, silent=True)
    assert os.listdir(out) == ['output.png']
    assert np.asarray(Image.open(out / 'output.png')).shape == (40, 48, 3)
    assert made == [(4, 240, 'nearest+conv', '3conv', 8, ('real_sr', '003_realSR_BSRGAN_DFOWMFC_s64w8_SwinIR-L_x4_GAN.pth'))]
    j = T.SwinIR('jpeg_compression_artifact_removal', '40', cache=sc)
    assert (j.tool_name, j.opt_task) == ('swinir_40', 'color_jpeg_car')
    assert T.SwinIR('denoising', '50', cache=sc).model_name == '005_colorDN_DFWB_s128w8_SwinIR-M_noise50.pth'
    assert '--large_model' in tool._get_cmd_opts() and '--large_model' not in j._get_cmd_opts()
    with pytest.raises(KeyError):
        T.SwinIR('denoising', '25')              # commented out in the reference (multitask_tools.py:106)


def test_service_api(cache):
    svc = S.RestormerService(task='Real_Denoising', weight_dir='/w', cache=cache)
    path, params = svc.get_weights_and_parameters('Gaussian_Gray_Denoising')   # model_serving.py:64-98
    assert path == '/w/gaussian_gray_denoising_blind.pth'
    assert params['inp_channels'] == 1 and params['out_channels'] == 1 and params['LayerNorm_type'] == 'BiasFree'
    assert svc.get_weights_and_parameters('Deraining')[1]['LayerNorm_type'] == 'WithBias'
    with pytest.raises(ValueError):
        svc.get_weights_and_parameters('nope')
    img = _img(17, 23)
    buf = io.BytesIO(); Image.fromarray(img).save(buf, format='PNG')
    out = svc.restore_image(buf.getvalue(), 'Real_Denoising', tile=64, tile_overlap=16)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(out))), 255 - img)
    assert FakeBackend.calls[-1] == ((17, 23, 3), 64, 16)
    assert list(svc.models.keys()) == ['Real_Denoising']
    # gray task decodes to one channel (model_serving.py:160-161)
    g = svc.restore_image(buf.getvalue(), 'Gaussian_Gray_Denoising')
    assert np.asarray(Image.open(io.BytesIO(g))).ndim == 2
    assert S.allowed_file('a.PNG') and not S.allowed_file('a.gif')


def test_restore_alias(cache):
    img = _img(9, 8)
    assert np.array_equal(S.restore(img, task='denoising', cache=cache), 255 - img)
    assert np.array_equal(S.restore(Image.fromarray(img), task='Deraining', cache=cache), 255 - img)
    with pytest.raises(ValueError):
        S.restore(img, task='colorize', cache=cache)
    with pytest.raises(ValueError):
        S.restore(img, tool='swinir', cache=cache)
    with pytest.raises(ValueError):
        S.restore(img, task='deraining', tool='swinir_gan', cache=cache)       # SwinIR has no deraining model

    class FakeSR:
        def __init__(self, cfg, sd, device):
            self.cfg = cfg

        def restore(self, im, tile=None, tile_overlap=32):
            return np.repeat(np.repeat(im, self.cfg.upscale, 0), self.cfg.upscale, 1)
    sc = T.SwinIRCache(state_dict_loader=lambda task, name, wd: {}, backend_factory=FakeSR)
    assert S.restore(img, task='super-resolution', tool='swinir_psnr', cache=sc).shape == (36, 32, 3)
    assert S.restore(img, task='denoising', tool='swinir_50', cache=sc).shape == (9, 8, 3)


def test_task_table_matches_reference():
    # demo.py:52-70: BiasFree for the three denoising tasks, 1 channel for gray
    for t in SUPPORTED_TASKS:
        cfg = task_config(t)
        assert (cfg.LayerNorm_type == 'BiasFree') == ('Denoising' in t)
        assert cfg.inp_channels == (1 if t == 'Gaussian_Gray_Denoising' else 3)
        assert (cfg.dim, cfg.num_blocks, cfg.heads, cfg.num_refinement_blocks, cfg.bias) == (48, [4, 6, 6, 8], [1, 2, 4, 8], 4, False)


def test_abi_library_exports_every_declared_symbol():
    """The C-ABI library loads here (no GPU) and exports every entry point include/ira.h declares."""
    from ira_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'ira.h')).read()
    declared = set(re.findall(r'\b(ira_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ira_restormer_config'}
    lib = _capi.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    # host-only entry point works without a device: the 4K plan of SURVEY 8a2
    assert _capi.tile_plan(2160, 3840, 512, 32) == (2160, 3840, 512, 512, 5, 8)
    assert _capi.tile_plan(75, 91, None, 32) == (80, 96, 80, 96, 1, 1)
    assert _capi.tile_plan(70, 100, 512, 32) == (72, 104, 72, 72, 1, 2)
    with pytest.raises(_capi.IraError):
        _capi.tile_plan(100, 100, 60, 32)          # "tile size should be multiple of 8" (demo.py:136)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from ira_amd._capi import IraError
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config
    from ira_amd.weights import seeded_state_dict
    cfg = reduced_config()
    with pytest.raises(IraError):
        RestormerBackend(cfg, seeded_state_dict(cfg, 0))


def test_swinir_service_api():
    """model_serving.py (SwinIR): define_model presets, the load_model cache key, process_image's BGR float contract."""
    from ira_amd import swinir_service as SS
    cfg, key = SS.define_model('real_sr', scale=4, large_model=True)
    assert (cfg.embed_dim, len(cfg.depths), cfg.upsampler, cfg.resi_connection, key) == (240, 9, 'nearest+conv', '3conv', 'params_ema')
    cfg, key = SS.define_model('color_jpeg_car')
    assert (cfg.window_size, cfg.img_range, cfg.img_size, key) == (7, 255., 126, 'params')
    assert SS.define_model('classical_sr', scale=2)[0].img_size == 48            # :45
    with pytest.raises(ValueError):
        SS.define_model('nope')

    class Fake:
        def __init__(self, cfg, sd, device):
            self.cfg, self.sd = cfg, sd

        def restore(self, img, tile=None, tile_overlap=32):
            s = self.cfg.upscale
            return np.repeat(np.repeat(img, s, 0), s, 1)
    svc = SS.SwinIRService(model_path='/w/x.pth', state_dict_loader=lambda path, key: (path, key), backend_factory=Fake)
    m, err = svc.load_model('real_sr', scale=4, large_model=True)
    assert err is None and m.sd == ('/w/x.pth', 'params_ema') and list(svc.models) == ['real_sr_4_15_40_True']
    assert svc.load_model('real_sr', scale=4, large_model=True)[0] is m
    bgr = _img(6, 5)[:, :, ::-1].astype(np.float32) / 255.
    out = svc.process_image(bgr, m, 4, 8)
    assert out.dtype == np.uint8 and out.shape == (24, 20, 3)
    assert np.array_equal(out[::4, ::4], np.rint(bgr * 255).astype(np.uint8))     # BGR in, BGR out
    with pytest.raises(ValueError):
        svc.process_image(bgr, m, 2, 8)
    miss = SS.SwinIRService(model_path='/nonexistent.pth')
    assert miss.load_model('color_dn')[1].startswith('Model file not found')


def test_xrestormer_tool_contract(tmp_path):
    """multitask_tools.py:66-78 + the subtask configs: name, checkpoint per subtask, untiled pad-to-64 harness call."""
    calls = []

    class FakeX:
        def __init__(self, cfg, sd, device):
            self.cfg, self.sd = cfg, sd

        def restore_padcrop(self, img, tile_size=None, tile_pad=32, padding_size=64):
            calls.append((img.shape, tile_size, tile_pad, padding_size))
            return 255 - img
    xc = T.XRestormerCache(state_dict_loader=lambda name, wd: name, backend_factory=FakeX)
    tool = T.XRestormer('dehazing', cache=xc)
    assert (tool.tool_name, tool.subtask, tool.model_name, tool.padding_size) == ('xrestormer', 'dehazing', 'dehaze_300k.pth', 64)
    inp, out = tmp_path / 'in', tmp_path / 'out'
    inp.mkdir(); out.mkdir()
    img = _img(21, 30)
    Image.fromarray(img).save(inp / 'input.png')
    tool(inp, out, silent=True)
    assert os.listdir(out) == ['output.png'] and np.array_equal(np.asarray(Image.open(out / 'output.png')), 255 - img)
    assert calls == [((21, 30, 3), None, 32, 64)]
    assert xc.get('dehaze_300k.pth').cfg.scale == 1 and xc.get('dehaze_300k.pth').cfg.spatial_heads == [1, 2, 4, 8] and xc.get('dehaze_300k.pth').sd == 'dehaze_300k.pth'
    assert T.XRestormer('deraining', cache=xc).model_name == 'derain_155k.pth'
    sr = T.XRestormer('super_resolution', cache=xc)
    assert (sr.model_name, sr.padding_size, sr.scale) == ('sr_300k.pth', 16, 4)
    with pytest.raises(KeyError):
        T.XRestormer('brightening', cache=xc)


def _empty(p):
    p.mkdir()
    return p


def test_mprnet_tool_contract(tmp_path):
    """multitask_tools.py:176-204 + MPRNet/demo.py:44-50,86-87: task option per subtask, jpg / png inputs only, '<stem>.png'."""
    calls = []

    class FakeBackend:
        def restore(self, img, tile=None):
            calls.append((img.shape, tile))
            return img

    loaded = []
    cache = T.MPRNetCache(state_dict_loader=lambda task, wd: loaded.append(task) or {}, backend_factory=lambda cfg, sd, dev: FakeBackend())
    tool = T.MPRNet('motion deblurring', cache)
    assert (tool.tool_name, tool.subtask, tool.opt_task) == ('mprnet', 'motion deblurring', 'Deblurring')
    assert T.MPRNet('denoising', cache).opt_task == 'Denoising' and T.MPRNet('deraining', cache).opt_task == 'Deraining'
    with pytest.raises(KeyError):
        T.MPRNet('dehazing', cache)
    inp = tmp_path / 'in'; inp.mkdir()
    out = tmp_path / 'out'; out.mkdir()
    Image.fromarray(_img(seed=1)).save(inp / 'a.png')
    tool(inp, out, silent=True)
    assert sorted(p.name for p in out.glob('*')) == ['output.png'] and len(calls) == 1 and calls[0][1] is None   # tool.py:56-62 renames a.png
    out_b = tmp_path / 'out_b'; out_b.mkdir()
    tool(inp, out_b, silent=True)
    assert loaded == ['Deblurring']                      # the network is built once per task, not per call
    other = tmp_path / 'other'; other.mkdir()
    (other / 'a.bmp').write_bytes(b'x')                  # demo.py globs jpg / png only
    with pytest.raises(Exception, match='No files found'):
        tool(other, _empty(tmp_path / 'out_c'), silent=True)
    from ira_amd.mprnet import task_config
    assert (task_config('Deraining').n_feat, task_config('Deraining').scale_orsnetfeats) == (40, 16)


def test_missing_rccl_is_an_error_not_a_crash(tmp_path):
    """with librccl.so not loadable the first ira_comm_* call must return IRA_ERR_COMM with a message (it used to
    build a std::string from a second, NULL dlerror() and crash).  IRA_RCCL_LIB names the one library the loader tries."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ira_amd import _capi\n"
        "lib = _capi.load_library()\n"
        "buf = (C.c_ubyte * 128)()\n"
        "rc = lib.ira_comm_unique_id(buf)\n"
        "lib.ira_last_error.restype = C.c_char_p\n"
        "msg = lib.ira_last_error(None).decode()\n"
        "print(rc, '|', msg)\n"
        "assert rc == 6, rc\n"
        "assert 'cannot load librccl.so' in msg and 'no-such-dir' in msg, msg\n"
        "rc2 = lib.ira_comm_unique_id(buf)\n"
        "assert rc2 == 6\n"
    ) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'image-restoration-agent_amd')
    env = dict(os.environ, IRA_RCCL_LIB=str(tmp_path / 'no-such-dir' / 'librccl.so'))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
