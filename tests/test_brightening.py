"""Brightening tools: oracle properties on the CPU, bit-exact HIP-vs-oracle on the GPU.  The oracle restates
OpenCV 4.8's 8-bit HSV transforms and CLAHE and is PARITY UNPINNED (see its header); the kernels must equal it byte for byte."""
import colorsys
from pathlib import Path

import numpy as np
import pytest

from oracle import brightening_oracle as B

TOOLS = ('constant_shift', 'gamma_correction', 'histogram_equalization')


def _img(h, w, seed, low_light=True):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([80 + 60 * np.sin(xx / 9.0), 90 + 70 * np.cos(yy / 7.0), 60 + 50 * np.sin((xx + yy) / 11.0)], -1)
    img = base * (0.35 if low_light else 1.0) + rng.normal(0, 6, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_v_luts_follow_the_reference_expressions():
    lut = B.v_lut('constant_shift')
    assert lut[0] == 40 and lut[215] == 255 and lut[214] == 254 and lut[255] == 255
    g = B.v_lut('gamma_correction')
    assert g[0] == 0 and g[255] == 255 and np.all(np.diff(g.astype(int)) >= 0) and g[64] == round(255 * (64 / 255) ** (1 / 1.5))
    x = np.power(np.arange(256) / 255.0, 1 / 1.5) * 255
    assert np.abs((x - np.floor(x)) - 0.5).min() > 1e-3          # no entry sits on a rounding boundary: cv2.pow's last bits cannot matter


def test_forward_hsv_is_the_integer_transform():
    img = _img(40, 50, 0, low_light=False)
    h, s, v = B.rgb_to_hsv_u8(img)
    assert h.max() < 180 and np.array_equal(v, img.max(-1))
    for y in range(0, 40, 3):
        for x in range(0, 50, 7):
            hh, ss, _ = colorsys.rgb_to_hsv(*(img[y, x] / 255.0))
            dh = abs(hh * 180 - int(h[y, x]))
            assert min(dh, 180 - dh) <= 1.01 and abs(ss * 255 - int(s[y, x])) <= 1.01
    gray = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, 2)
    h, s, v = B.rgb_to_hsv_u8(gray)
    assert not h.any() and not s.any()
    assert np.array_equal(B.hsv_to_rgb_u8(h, s, v), gray) and np.array_equal(B.hsv_to_rgb_u8(h, s, v, 'scalar'), gray)


def test_two_opencv_formulations_differ_by_at_most_one_lsb():
    img = _img(96, 128, 1, low_light=False)
    for t in TOOLS:
        a, b = B.brighten(img, t, 'simd'), B.brighten(img, t, 'scalar')
        d = np.abs(a.astype(int) - b)
        assert d.max() <= 1 and (d > 0).mean() < 1e-2


def test_round_trip_is_close_and_tools_brighten():
    img = _img(64, 72, 2)
    h, s, v = B.rgb_to_hsv_u8(img)
    assert np.abs(B.hsv_to_rgb_u8(h, s, v).astype(int) - img).max() <= 6          # 8-bit H and S quantisation
    for t in TOOLS:
        out = B.brighten(img, t)
        assert out.shape == img.shape and out.dtype == np.uint8 and out.mean() > img.mean() + 5


def test_clahe_flat_plane_and_padding_rule():
    flat = np.full((64, 64), 77, np.uint8)
    luts, th, tw = B.clahe_luts(flat)
    assert (th, tw) == (8, 8) and luts.shape == (64, 256)
    out = B.clahe_u8(flat)
    assert len(np.unique(out)) == 1                                            # every tile has the same look-up
    _, th, tw = B.clahe_luts(np.zeros((64, 70), np.uint8))
    assert (th, tw) == (9, 9)                                                  # 64 % 8 == 0 still gets 8 more rows when the width needs padding
    v = _img(61, 83, 3)[..., 0]
    out = B.clahe_u8(v)
    assert out.shape == v.shape and out.std() > v.std()


def test_unknown_tool_is_an_error():
    with pytest.raises(ValueError):
        B.v_lut('maxim')
    from ira_amd._capi import BRIGHTEN_TOOLS
    assert set(BRIGHTEN_TOOLS) == set(TOOLS)


def test_toolbox_mirrors_the_reference_registration():
    from ira_amd.tools import brightening_toolbox, build_executor
    names = [t.tool_name for t in brightening_toolbox()]
    assert names == ['histogram_equalization', 'gamma_correction', 'constant_shift']       # brightening/__init__.py:68-73
    assert all(t.subtask == 'brightening' for t in brightening_toolbox())
    assert 'brightening' in build_executor().subtasks


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('tool', TOOLS)
@pytest.mark.parametrize('h,w,low', [(64, 64, True), (61, 83, True), (37, 50, False), (128, 200, True), (16, 19, False), (240, 135, True)])
def test_gpu_brighten_is_bit_exact(tool, h, w, low):
    from ira_amd._capi import brighten_u8
    img = _img(h, w, h + w, low)
    got = brighten_u8(img, tool)
    ref = B.brighten(img, tool)
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.array_equal(got, ref), (np.abs(got.astype(int) - ref).max(), (got != ref).mean())


@pytest.mark.gpu
def test_gpu_brighten_all_colours_and_device_entry():
    """Every (H sector, S, V) corner: a 256 x 256 x 3 sweep of the colour cube's faces plus random pixels; and the device entry point."""
    import ctypes as C
    import torch
    from ira_amd._capi import BRIGHTEN_TOOLS, brighten_u8, check, load_library
    rng = np.random.default_rng(5)
    a = np.arange(256, dtype=np.uint8)
    faces = [np.stack(np.broadcast_arrays(a[:, None], a[None, :], np.uint8(c)), -1) for c in (0, 128, 255)]
    img = np.concatenate([np.roll(f, k, -1) for f in faces for k in range(3)] + [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)], 0)
    for tool in ('constant_shift', 'gamma_correction'):
        assert np.array_equal(brighten_u8(img, tool), B.brighten(img, tool))
    lib = load_library()
    x = torch.from_numpy(img).cuda()
    y = torch.empty_like(x)
    for tool in TOOLS:
        check(lib.ira_brighten_u8_device(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), BRIGHTEN_TOOLS[tool], x.data_ptr(),
                                         img.shape[0], img.shape[1], y.data_ptr()), None, 'ira_brighten_u8_device')
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), B.brighten(img, tool))


@pytest.mark.gpu
def test_gpu_brightening_tool_directory_protocol(tmp_path):
    from ira_amd.tools import GammaCorrection, read_image, write_png
    img = _img(50, 70, 9)
    (tmp_path / 'in').mkdir(); (tmp_path / 'out').mkdir()
    write_png(tmp_path / 'in' / 'input.png', img)
    GammaCorrection()(tmp_path / 'in', tmp_path / 'out', silent=True)
    assert [p.name for p in (tmp_path / 'out').glob('*')] == ['output.png']
    assert np.array_equal(read_image(tmp_path / 'out' / 'output.png'), B.brighten(img, 'gamma_correction'))


@pytest.mark.gpu
def test_gpu_brighten_argument_checks():
    from ira_amd._capi import IraError, brighten_u8
    with pytest.raises(IraError, match='16 x 16'):
        brighten_u8(np.zeros((8, 40, 3), np.uint8), 'histogram_equalization')
    with pytest.raises(KeyError):
        brighten_u8(np.zeros((8, 8, 3), np.uint8), 'maxim')
