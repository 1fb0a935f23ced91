"""Agent-facing tool / registry API, source-compatible with the reference's

    AgenticIR/executor/tool.py:8-90            class Tool
    AgenticIR/executor/multitask_tools.py:136-173   class Restormer(Tool);  :81-133 class SwinIR(Tool)
    AgenticIR/executor/__init__.py:24-169      class Executor, singleton ``executor``

with the one difference that matters: ``Restormer._invoke`` runs the in-process HIP backend instead of
``subprocess.run("conda run -n restormer python demo.py ...")`` (tool.py:64-70), so there is no process
start, no torch import and no weight reload per call (SURVEY section 1, L2 -> L1 boundary).

Contract kept (tool.py:33-62): ``tool(input_dir, output_dir, silent=False)``; ``input_dir`` holds exactly
one image; ``output_dir`` is empty before and holds exactly ``output.png`` after; failures are Python
exceptions (the agent catches and prints them, agentic_api.py:191-198).  Attributes read by the agent:
``tool_name`` and ``subtask`` (utils/util.py:337-338, agentic_api.py:189,202).
"""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Callable, Optional

import numpy as np

from .config import SUBTASK_TO_TASK, TASK_WEIGHT_FILE, canonical_subtask, task_config


class Tool:
    """Abstract tool (tool.py:8-31): same constructor arguments and attributes as the reference."""

    def __init__(self, tool_name: str, subtask: str, work_dir: Optional[Path] = None,
                 script_rel_path: Optional[Path | str] = None):
        self.tool_name: str = tool_name
        self.subtask: str = subtask
        self.work_dir: Optional[Path] = None
        self.script_path: Optional[Path] = None
        if work_dir is not None:
            assert script_rel_path is not None, "If `work_dir` is provided, `script_rel_path` should also be provided."
            self.work_dir = Path().resolve() / 'executor' / subtask / 'tools' / work_dir
            self.script_path = self.work_dir / script_rel_path

    def __call__(self, input_dir: Path, output_dir: Path, silent: bool = False, *args) -> None:
        """tool.py:33-49, print-for-print."""
        input_dir, output_dir = Path(input_dir), Path(output_dir)
        if not silent:
            print('-' * 100)
            print(f"Subtask\t: {self.subtask}")
            print(f"Tool\t: {self.tool_name}")
            print(f"Input\t: {list(input_dir.glob('*'))[0]}")
        start_time = time.time()
        self.input_dir = input_dir
        self.output_dir = output_dir
        self._precheck()
        self._invoke(*args)
        self._postcheck()
        end_time = time.time()
        if not silent:
            print(f"Output\t: {list(output_dir.glob('*'))[0]}")
            print(f"Time\t: {round(end_time - start_time, 3)}s")

    def _precheck(self) -> None:
        """tool.py:51-54."""
        assert len(os.listdir(self.input_dir)) == 1, "The input directory should contain the input only."
        assert os.listdir(self.output_dir) == [], "The output directory should be empty."

    def _postcheck(self) -> None:
        """tool.py:56-62: exactly one output, renamed to ``output.png`` if needed."""
        output = list(self.output_dir.glob('*'))
        assert len(output) == 1, "There're other files in the same directory as the output image."
        if output[0].name != 'output.png':
            output[0].replace(self.output_dir / 'output.png')

    def _invoke(self, *args) -> None:
        raise NotImplementedError("only in-process HIP tools are provided by ira_amd (tool.py:64-70 runs a subprocess)")

    def _preprocess(self):
        pass

    def _postprocess(self):
        pass


# ------------------------------------------------------------------------------------------------
# weights: where a ``.pth`` comes from.  The reference reads
#   <work_dir>/<Task dir>/pretrained_models/<file>.pth            (demo.py:52-69)
#   <root>/<weight_dir>/<file>.pth                                (model_serving.py:52-53,77-94)
# Neither exists offline, so the loader is injectable; the default looks under $IRA_RESTORMER_WEIGHTS.

_TASK_DIR = {   # demo.py:53-66
    'Motion_Deblurring': 'Motion_Deblurring', 'Single_Image_Defocus_Deblurring': 'Defocus_Deblurring',
    'Deraining': 'Deraining', 'Real_Denoising': 'Denoising', 'Gaussian_Color_Denoising': 'Denoising',
    'Gaussian_Gray_Denoising': 'Denoising',
}


def default_state_dict_loader(task: str, weight_dir: Optional[str] = None):
    """``torch.load(weights)['params']`` (demo.py:101-102).  PyTorch is used for nothing else."""
    root = weight_dir or os.environ.get('IRA_RESTORMER_WEIGHTS', '')
    fname = TASK_WEIGHT_FILE[task]
    candidates = [os.path.join(root, fname), os.path.join(root, _TASK_DIR[task], 'pretrained_models', fname)]
    for path in candidates:
        if root and os.path.isfile(path):
            import torch
            ckpt = torch.load(path, map_location='cpu')
            return ckpt['params'] if 'params' in ckpt else ckpt
    raise FileNotFoundError(f"Restormer weights for task {task!r} not found (looked for {candidates}); set "
                            "IRA_RESTORMER_WEIGHTS or pass weight_dir / state_dict_loader")


def seeded_state_dict_loader(seed: int = 0) -> Callable:
    """Loader used by tests / benchmarks: the build's deterministic weights (no checkpoint offline)."""
    def load(task: str, weight_dir=None):
        from .weights import seeded_state_dict
        return seeded_state_dict(task_config(task), seed)
    return load


class ModelCache:
    """key -> backend on one device, built on first use and kept for the life of the process; plays the role of
    RestormerService.models (model_serving.py:57,101-105: 'model has been loaded, use cache').  One implementation for every
    model family: a subclass only says how a key turns into (config, state dict) and which backend class holds them."""
    backend_name = ''                      # class in ira_amd.backend
    default_loader: Optional[Callable] = None

    def __init__(self, device: int = 0, state_dict_loader: Optional[Callable] = None, weight_dir: Optional[str] = None,
                 backend_factory: Optional[Callable] = None):
        self.device, self.weight_dir = device, weight_dir
        self.loader = state_dict_loader or type(self).default_loader
        self.factory = backend_factory
        self.models: dict = {}

    def _get(self, key, make_cfg: Callable, load_sd: Callable):
        if key not in self.models:
            cfg, sd = make_cfg(), load_sd()
            if self.factory is not None:
                self.models[key] = self.factory(cfg, sd, self.device)
            else:
                from . import backend
                self.models[key] = getattr(backend, self.backend_name)(cfg, sd, device=self.device)
        return self.models[key]

    def close(self):
        for be in self.models.values():
            if hasattr(be, 'close'):
                be.close()
        self.models.clear()

    @classmethod
    def default(cls):
        """The process-wide cache of this family on $IRA_DEVICE / $LOCAL_RANK (created on first use)."""
        if _DEFAULTS.get(cls) is None:
            _DEFAULTS[cls] = cls(device=_default_device())
        return _DEFAULTS[cls]


_DEFAULTS: dict = {}


def _default_device() -> int:
    return int(os.environ.get('IRA_DEVICE', os.environ.get('LOCAL_RANK', '0')))


class BackendCache(ModelCache):
    """task -> RestormerBackend."""
    backend_name = 'RestormerBackend'
    default_loader = staticmethod(default_state_dict_loader)

    def get(self, task: str):
        return self._get(task, lambda: task_config(task), lambda: self.loader(task, self.weight_dir))


def default_cache() -> BackendCache:
    return BackendCache.default()


def set_default_cache(cache: Optional[BackendCache]):
    _DEFAULTS[BackendCache] = cache


# ------------------------------------------------------------------------------------------------
def decode_image(data: bytes, gray: bool = False) -> np.ndarray:
    """cv2.imdecode / cv2.imread of the reference (demo.py:40-47: IMREAD_COLOR + BGR2RGB, or IMREAD_GRAYSCALE; model_serving.py:126-134):
    uint8 (h,w,3) RGB or, for the grey task, (h,w,1).  PNG streams are decoded by the library (csrc/ira_png.cpp); other containers (the
    reference's demo inputs are .jpg) and the PNG variants the native decoder declines (16-bit, interlaced, 1/2/4-bit) go to PIL.  Alpha
    is dropped and grey is replicated as IMREAD_COLOR does; RGB -> grey uses PIL's 'L' weights (19595 R + 38470 G + 7471 B + 32768) >> 16,
    which is what this wrapper has always returned."""
    from ._capi import IraError, decode_png
    a = None
    if data[:8] == b'\x89PNG\r\n\x1a\n':
        try:
            a = decode_png(data)
        except IraError:
            a = None                                     # a PNG variant left to PIL (or a broken stream: PIL raises its own error)
    if a is None:
        import io
        from PIL import Image
        with Image.open(io.BytesIO(data)) as im:
            if gray:
                return np.asarray(im.convert('L'), dtype=np.uint8)[:, :, None].copy()
            return np.asarray(im.convert('RGB'), dtype=np.uint8).copy()
    c = a.shape[2]
    if c in (2, 4):
        a = a[:, :, :c - 1]                              # alpha is ignored (IMREAD_COLOR / IMREAD_GRAYSCALE)
        c -= 1
    if gray:
        if c == 1:
            return np.ascontiguousarray(a)
        r, g, b = (a[:, :, i].astype(np.uint32) for i in range(3))
        return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)[:, :, None]
    return np.ascontiguousarray(a) if c == 3 else np.repeat(a, 3, axis=2)


def read_image(path, gray: bool = False) -> np.ndarray:
    """demo.py:40-47 on a file: see decode_image."""
    with open(path, 'rb') as f:
        return decode_image(f.read(), gray)


def encode_png(img: np.ndarray, level: int = 1) -> bytes:
    """Lossless 8-bit PNG (gray or RGB) with filter type 0 on every row and one zlib stream.  PIL's / libpng's adaptive row
    filtering costs 1.4 s for a 3840x2160 frame whatever the zlib level; this is 0.15 s, and any PNG reader decodes it."""
    import struct
    import zlib
    a = np.ascontiguousarray(img)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("encode_png expects uint8 (h,w), (h,w,1) or (h,w,3)")
    h, w = a.shape[:2]
    rows = np.zeros((h, 1 + (w * (3 if a.ndim == 3 else 1))), np.uint8)
    rows[:, 1:] = a.reshape(h, -1)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    ihdr = struct.pack('>IIBBBBB', w, h, 8, 2 if a.ndim == 3 else 0, 0, 0, 0)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', ihdr) + chunk(b'IDAT', _deflate_parallel(rows.tobytes(), level)) + chunk(b'IEND', b'')


def _deflate_parallel(raw: bytes, level: int, stripe: int = 1 << 20) -> bytes:
    """One valid zlib stream built from independently deflated 1 MB stripes (pigz's scheme: every stripe but the last ends with a
    sync flush, so the raw deflate blocks concatenate; the Adler-32 trailer is computed over the whole input).  zlib releases
    the GIL, so the stripes run on a thread pool: 25 MB in ~0.1 s instead of 0.75 s."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    n = max(1, (len(raw) + stripe - 1) // stripe)
    if level > 0 and len(raw) > (1 << 20):
        mid = len(raw) // 2
        probe = raw[mid:mid + (1 << 18)]
        if len(zlib.compress(probe, 1)) > 0.9 * len(probe):
            level = 0                                  # noise-like content does not deflate: store it (40 ms instead of 0.8 s for a 4K frame)

    def work(i):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = co.compress(raw[i * stripe:(i + 1) * stripe])
        return data + (co.flush(zlib.Z_FINISH) if i == n - 1 else co.flush(zlib.Z_SYNC_FLUSH))
    with ThreadPoolExecutor(max_workers=min(16, n)) as ex:
        parts = list(ex.map(work, range(n)))
    return b'\x78\x01' + b''.join(parts) + struct.pack('>I', zlib.adler32(raw) & 0xFFFFFFFF)


def write_png(path, img: np.ndarray):
    """demo.py:43-50 (cv2.imwrite PNG): lossless, so the encoder does not matter to the pixels."""
    with open(path, 'wb') as f:
        f.write(encode_png(img))


class Restormer(Tool):
    """Restormer tool for denoising, motion deblurring, defocus deblurring and deraining
    (multitask_tools.py:136-173), running on the MI355X backend in-process.

    Like the reference tool it passes no ``--tile`` (multitask_tools.py:159-165): whole-image mode
    (demo.py:129-131).  ``tile`` / ``tile_overlap`` can be set for the tiled mode of demo.py:132-153.
    """

    def __init__(self, subtask: str, cache: Optional[BackendCache] = None, tile: Optional[int] = None,
                 tile_overlap: int = 32):
        super().__init__(tool_name='restormer', subtask=subtask, work_dir='Restormer', script_rel_path='demo.py')
        opt_dict = SUBTASK_TO_TASK      # multitask_tools.py:151-156
        self.opt_task = opt_dict[canonical_subtask(subtask)]
        self._cache = cache
        self.tile, self.tile_overlap = tile, tile_overlap

    def _get_cmd_opts(self) -> list:
        """Kept for introspection parity (multitask_tools.py:159-165); not used to spawn anything."""
        return ["--task", self.opt_task, "--input_dir", self.input_dir, "--result_dir", self.output_dir]

    def _invoke(self, *args) -> None:
        self._preprocess()
        cache = self._cache or default_cache()
        be = cache.get(self.opt_task)
        src = sorted(Path(self.input_dir).glob('*'))[0]
        gray = self.opt_task == 'Gaussian_Gray_Denoising'
        img = read_image(src, gray=gray)
        out = be.restore(img, tile=self.tile, tile_overlap=self.tile_overlap)
        # demo.py writes {result_dir}/{task}/{stem}.png and multitask_tools.py:167-173 moves it to output.png
        write_png(Path(self.output_dir) / 'output.png', out)
        self._postprocess()


# SwinIR tool: subtask -> (inference.py --task, {pretrained_on: checkpoint})   (multitask_tools.py:97-116)
SWINIR_OPTS = {
    'super_resolution': ('real_sr', {'gan': '003_realSR_BSRGAN_DFOWMFC_s64w8_SwinIR-L_x4_GAN.pth',
                                     'psnr': '003_realSR_BSRGAN_DFOWMFC_s64w8_SwinIR-L_x4_PSNR.pth'}),
    'denoising': ('color_dn', {'15': '005_colorDN_DFWB_s128w8_SwinIR-M_noise15.pth',
                               '50': '005_colorDN_DFWB_s128w8_SwinIR-M_noise50.pth'}),
    'jpeg_compression_artifact_removal': ('color_jpeg_car', {'40': '006_colorCAR_DFWB_s126w7_SwinIR-M_jpeg40.pth'}),
}


def default_swinir_loader(task: str, model_name: str, weight_dir: Optional[str] = None):
    """``torch.load(model_path)[param_key_g]`` of SwinIR/inference.py:202-203 ('params_ema' for real_sr, :161).  Looks in
    ``weight_dir`` or $IRA_SWINIR_WEIGHTS (the reference's ``SwinIR/model_zoo/swinir`` directory, multitask_tools.py:124)."""
    from .swinir import PARAM_KEY
    root = weight_dir or os.environ.get('IRA_SWINIR_WEIGHTS', '')
    path = os.path.join(root, model_name)
    if not (root and os.path.isfile(path)):
        raise FileNotFoundError(f"SwinIR weights {model_name!r} not found in {root!r}; set IRA_SWINIR_WEIGHTS or pass a loader")
    import torch
    ckpt = torch.load(path, map_location='cpu')
    key = PARAM_KEY.get(task, 'params')
    return ckpt[key] if key in ckpt else ckpt


class SwinIRCache(ModelCache):
    """(task, checkpoint name) -> SwinIRBackend."""
    backend_name = 'SwinIRBackend'
    default_loader = staticmethod(default_swinir_loader)

    def get(self, task: str, model_name: str):
        def cfg():
            from .swinir import task_config as swinir_task_config
            return swinir_task_config(task, scale=4, large_model=True)          # multitask_tools.py:128-132: --scale 4 --large_model
        return self._get((task, model_name), cfg, lambda: self.loader(task, model_name, self.weight_dir))


def default_swinir_cache() -> SwinIRCache:
    return SwinIRCache.default()


class SwinIR(Tool):
    """SwinIR tool for real SR (x4, SwinIR-L), colour denoising and JPEG artifact removal (multitask_tools.py:81-133),
    in-process on the MI355X backend.  Like the reference tool it passes no ``--tile``: whole-image mode
    (inference.py:295-297); ``tile`` / ``tile_overlap`` select the tiled mode of inference.py:298-320."""

    def __init__(self, subtask: str, pretrained_on: str, cache: Optional[SwinIRCache] = None, tile: Optional[int] = None,
                 tile_overlap: int = 32):
        super().__init__(tool_name=f"swinir_{pretrained_on}", subtask=subtask, work_dir="SwinIR", script_rel_path="inference.py")
        self.opt_task, opt_model_names = SWINIR_OPTS[canonical_subtask(subtask)]
        self.model_name = opt_model_names[pretrained_on]
        self._cache = cache
        self.tile, self.tile_overlap = tile, tile_overlap

    def _get_cmd_opts(self) -> list:
        """Kept for introspection parity (multitask_tools.py:118-133); not used to spawn anything."""
        opts = ["--task", self.opt_task, "--model_path", f"SwinIR/model_zoo/swinir/{self.model_name}",
                "--folder_lq", getattr(self, 'input_dir', None), "--save_dir", getattr(self, 'output_dir', None)]
        if canonical_subtask(self.subtask) == "super_resolution":
            opts += ['--scale', '4', '--large_model']
        return opts

    def _invoke(self, *args) -> None:
        self._preprocess()
        be = (self._cache or default_swinir_cache()).get(self.opt_task, self.model_name)
        src = sorted(Path(self.input_dir).glob('*'))[0]
        img = read_image(src, gray=False)                    # the agent's three SwinIR tasks are colour (inference.py:236-260)
        out = be.restore(img, tile=self.tile, tile_overlap=self.tile_overlap)
        write_png(Path(self.output_dir) / 'output.png', out)   # inference.py:90 writes {imgname}.png; the tool renames it
        self._postprocess()


# X-Restormer tool: subtask -> (checkpoint, padding_size, scale) of executor/<subtask>/configs/xrestormer.yml
XRESTORMER_OPTS = {
    'denoising': ('denoise_300k.pth', 64, 1), 'dehazing': ('dehaze_300k.pth', 64, 1),
    'motion_deblurring': ('deblur_300k.pth', 64, 1), 'deraining': ('derain_155k.pth', 64, 1),
    'super_resolution': ('sr_300k.pth', 16, 4),
}


def default_xrestormer_loader(model_name: str, weight_dir: Optional[str] = None):
    """basicsr's load_network(strict) of ``path.pretrain_network_g`` (param key 'params' when present).  Looks in ``weight_dir``
    or $IRA_XRESTORMER_WEIGHTS (the reference's X-Restormer/experiments/pretrained directory)."""
    root = weight_dir or os.environ.get('IRA_XRESTORMER_WEIGHTS', '')
    path = os.path.join(root, model_name)
    if not (root and os.path.isfile(path)):
        raise FileNotFoundError(f"X-Restormer weights {model_name!r} not found in {root!r}; set IRA_XRESTORMER_WEIGHTS or pass a loader")
    import torch
    ckpt = torch.load(path, map_location='cpu')
    for key in ('params_ema', 'params'):
        if key in ckpt:
            return ckpt[key]
    return ckpt


class XRestormerCache(ModelCache):
    """checkpoint name -> XRestormerBackend (every preset shares one architecture)."""
    backend_name = 'XRestormerBackend'
    default_loader = staticmethod(default_xrestormer_loader)

    def get(self, model_name: str, scale: int = 1):
        def cfg():
            from .xrestormer import XRestormerConfig
            return XRestormerConfig(scale=scale)              # network_g of the configs: identical apart from scale
        return self._get(model_name, cfg, lambda: self.loader(model_name, self.weight_dir))


def default_xrestormer_cache() -> XRestormerCache:
    return XRestormerCache.default()


class XRestormer(Tool):
    """X-Restormer tool for denoising, dehazing, motion deblurring and deraining (multitask_tools.py:66-78 on the BasicSRModel
    base :8-63), in-process.  The reference writes a BasicSR config and spawns ``inference.py -opt cfg.yml``; the config has no
    ``tile`` section, so the image is reflect-padded to a multiple of 64 and run whole (xrestormer_model.py:53-63,138-150)."""

    def __init__(self, subtask: str, cache: Optional[XRestormerCache] = None, tile_size: Optional[int] = None, tile_pad: int = 32):
        super().__init__(tool_name="xrestormer", subtask=subtask, work_dir="X-Restormer", script_rel_path=Path("xrestormer") / 'inference.py')
        sub = canonical_subtask(subtask)
        if sub not in XRESTORMER_OPTS:
            raise KeyError(f"X-Restormer handles {sorted(XRESTORMER_OPTS)}, not {subtask!r}")
        self.model_name, self.padding_size, self.scale = XRESTORMER_OPTS[sub]
        self._cache = cache
        self.tile_size, self.tile_pad = tile_size, tile_pad

    def _invoke(self, *args) -> None:
        be = (self._cache or default_xrestormer_cache()).get(self.model_name, self.scale)
        src = sorted(Path(self.input_dir).glob('*'))[0]
        out = be.restore_padcrop(read_image(src, gray=False), tile_size=self.tile_size, tile_pad=self.tile_pad, padding_size=self.padding_size)
        write_png(Path(self.output_dir) / 'output.png', out)


class Executor:
    """Registry of toolboxes (executor/__init__.py:24-94,161-169)."""

    def __init__(self):
        self.toolbox_router: dict = {}
        self._executed_subtask_cnt: int = 0

    def register_subtask(self, subtask_name, subtask_toolbox) -> None:
        self.toolbox_router[subtask_name] = subtask_toolbox

    @property
    def subtasks(self) -> set:
        return set(self.toolbox_router.keys())

    @property
    def executed_subtask_cnt(self) -> int:
        return self._executed_subtask_cnt

    def execute_subtask(self, subtask: str, input_path: Path) -> Path:
        """executor/__init__.py:40-85: runs every tool of the subtask, directory protocol unchanged."""
        input_path = Path(input_path)
        self._executed_subtask_cnt += 1
        subtask_dir = input_path.parents[1] / f"subtask{self._executed_subtask_cnt}-{subtask.replace(' ', '_')}"
        subtask_dir.mkdir()
        toolbox = self.toolbox_router[subtask]
        output_path = None
        for tool_idx, tool in enumerate(toolbox, start=1):
            tool_dir = subtask_dir / f'tool{tool_idx}-{tool.tool_name}'
            output_dir = tool_dir / '0-img'
            output_dir.mkdir(parents=True)
            tool(input_dir=input_path.parent, output_dir=output_dir)
            output_path = list(output_dir.glob('*'))[0]
        return output_path

    def invoke_a_tool(self, subtask_name: str, tool_name: str, input_dir: Path, output_dir: Path):
        """executor/__init__.py:87-94."""
        for tool in self.toolbox_router[subtask_name]:
            if tool.tool_name == tool_name:
                tool(Path(input_dir), Path(output_dir))
                return


def default_mprnet_loader(task: str, weight_dir: Optional[str] = None):
    """MPRNet/demo.py:25-35,60: ``{task}/pretrained_models/model_{task.lower()}.pth``, key 'state_dict', optional 'module.' prefix.
    Looks in ``weight_dir`` or $IRA_MPRNET_WEIGHTS (the reference's MPRNet directory)."""
    from .mprnet import TASK_WEIGHT_FILE
    root = weight_dir or os.environ.get('IRA_MPRNET_WEIGHTS', '')
    path = os.path.join(root, TASK_WEIGHT_FILE[task])
    if not (root and os.path.isfile(path)):
        raise FileNotFoundError(f"MPRNet weights {TASK_WEIGHT_FILE[task]!r} not found in {root!r}; set IRA_MPRNET_WEIGHTS or pass a loader")
    import torch
    sd = torch.load(path, map_location='cpu')['state_dict']
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


class MPRNetCache(ModelCache):
    """task ('Denoising' | 'Deblurring' | 'Deraining') -> MPRNetBackend."""
    backend_name = 'MPRNetBackend'
    default_loader = staticmethod(default_mprnet_loader)

    def __init__(self, device: int = 0, state_dict_loader: Optional[Callable] = None, weight_dir: Optional[str] = None,
                 backend_factory: Optional[Callable] = None, config_fn: Optional[Callable] = None):
        super().__init__(device, state_dict_loader, weight_dir, backend_factory)
        self.config_fn = config_fn

    def get(self, task: str):
        def cfg():
            from .mprnet import task_config as mprnet_task_config
            return (self.config_fn or mprnet_task_config)(task)
        return self._get(task, cfg, lambda: self.loader(task, self.weight_dir))


def default_mprnet_cache() -> MPRNetCache:
    return MPRNetCache.default()


class MPRNet(Tool):
    """MPRNet tool for denoising, motion deblurring and deraining (multitask_tools.py:176-204), in-process.  The reference spawns
    ``demo.py --task T`` which loads the task's network per call and runs every image whole (demo.py:64-87)."""

    def __init__(self, subtask: str, cache: Optional[MPRNetCache] = None):
        super().__init__(tool_name='mprnet', subtask=subtask, work_dir='MPRNet', script_rel_path='demo.py')
        from .mprnet import SUBTASK_TO_TASK
        sub = canonical_subtask(subtask)
        if sub not in SUBTASK_TO_TASK:
            raise KeyError(subtask)                          # the reference's opt_dict[subtask] (:191-196)
        self.opt_task = SUBTASK_TO_TASK[sub]
        self._cache = cache

    def _get_cmd_opts(self) -> list:
        return ["--task", self.opt_task, "--input_dir", self.input_dir, "--result_dir", self.output_dir]

    def _invoke(self, *args) -> None:
        be = (self._cache or default_mprnet_cache()).get(self.opt_task)
        files = sorted(p for p in Path(self.input_dir).glob('*') if p.suffix in ('.jpg', '.JPG', '.png', '.PNG'))   # demo.py:44-47
        if not files:
            raise Exception(f"No files found at {self.input_dir}")                                                  # demo.py:49-50
        for src in files:
            out = be.restore(read_image(src, gray=False), tile=None)
            write_png(Path(self.output_dir) / (src.stem + '.png'), out)                                             # demo.py:86-87


# ------------------------------------------------------------------------------------------------
# brightening tools (AgenticIR/executor/brightening/__init__.py): HSV look-up on V, no weights
class BrighteningTool(Tool):
    """brightening/__init__.py:12-34: read the one input, RGB -> HSV, update V, HSV -> RGB, write ``output.png``.  The three cv2 calls
    between imread and imwrite are one ira_brighten_u8 call (HIP kernels; no CPU path)."""

    def __init__(self, tool_name: str, device: int = 0):
        super().__init__(tool_name=tool_name, subtask="brightening")
        self.device = device

    def _invoke(self) -> None:
        from ._capi import brighten_u8
        input_path = list(self.input_dir.glob('*'))[0]
        img = read_image(input_path)
        write_png(self.output_dir / 'output.png', brighten_u8(img, self.tool_name, self.device))


class ConstantShift(BrighteningTool):
    """:37-44, V + 40 clipped to 255."""

    def __init__(self, device: int = 0):
        super().__init__(tool_name="constant_shift", device=device)


class GammaCorrection(BrighteningTool):
    """:47-54, 255 (V / 255) ** (1 / 1.5)."""

    def __init__(self, device: int = 0):
        super().__init__(tool_name="gamma_correction", device=device)


class HistogramEqualization(BrighteningTool):
    """:57-64, cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(V)."""

    def __init__(self, device: int = 0):
        super().__init__(tool_name="histogram_equalization", device=device)


def brightening_toolbox(device: int = 0) -> list:
    """brightening/__init__.py:68-73, the reference's order (MAXIM is commented out there)."""
    return [HistogramEqualization(device), GammaCorrection(device), ConstantShift(device)]


# subtask keys exactly as the reference registers them (executor/__init__.py:162-169)
RESTORMER_SUBTASKS = {
    'denoising': 'denoising',
    'motion deblurring': 'motion_deblurring',
    'defocus deblurring': 'defocus_deblurring',
    'deraining': 'deraining',
}


def build_executor(cache: Optional[BackendCache] = None, swinir_cache: Optional[SwinIRCache] = None,
                   xrestormer_cache: Optional[XRestormerCache] = None, mprnet_cache: Optional[MPRNetCache] = None) -> Executor:
    """The tools this backend provides, under the reference's subtask keys and in the reference's toolbox order with the tools this
    backend does not build (MAXIM, DiffBIR, HAT, FBCNN, DRBNet, RIDCP, DehazeFormer) left out: denoising/__init__.py:12-19,
    motion_deblurring/__init__.py:10-15, defocus_deblurring/__init__.py:128-131, deraining/__init__.py:10-15, dehazing/__init__.py:51-56,
    super_resolution/__init__.py:50-56, jpeg_compression_artifact_removal/__init__.py:34-39, brightening/__init__.py:68-73 (complete)."""
    R = lambda sub: Restormer(subtask=sub, cache=cache)
    X = lambda sub: XRestormer(sub, xrestormer_cache)
    M = lambda sub: MPRNet(sub, mprnet_cache)
    S = lambda sub, on: SwinIR(sub, on, swinir_cache)
    ex = Executor()
    ex.register_subtask('super-resolution', [X('super_resolution'), S('super_resolution', 'gan'), S('super_resolution', 'psnr')])
    ex.register_subtask('denoising', [X('denoising'), S('denoising', '15'), S('denoising', '50'), M('denoising'), R('denoising')])
    ex.register_subtask('motion deblurring', [R('motion_deblurring'), M('motion_deblurring'), X('motion_deblurring')])
    ex.register_subtask('defocus deblurring', [R('defocus_deblurring')])
    ex.register_subtask('deraining', [X('deraining'), R('deraining'), M('deraining')])
    ex.register_subtask('dehazing', [X('dehazing')])
    ex.register_subtask('jpeg compression artifact removal', [S('jpeg_compression_artifact_removal', '40')])
    ex.register_subtask('brightening', brightening_toolbox(cache.device if cache is not None else _default_device()))
    return ex


def install_into(reference_executor, cache: Optional[BackendCache] = None, swinir_cache: Optional[SwinIRCache] = None,
                 xrestormer_cache: Optional[XRestormerCache] = None, mprnet_cache: Optional[MPRNetCache] = None) -> int:
    """Drop-in: replace every tool named 'restormer', 'swinir_*', 'xrestormer', 'mprnet' or one of the three brightening tools in an existing
    (reference) executor's toolboxes by the HIP-backed tool, leaving all other tools and the LLM orchestration untouched.
    Returns the count."""
    n = 0
    for key, toolbox in reference_executor.toolbox_router.items():
        for i, tool in enumerate(toolbox):
            name = getattr(tool, 'tool_name', None) or ''
            sub = canonical_subtask(getattr(tool, 'subtask', key))
            if name == 'restormer':
                toolbox[i] = Restormer(subtask=sub, cache=cache)
                n += 1
            elif name == 'xrestormer' and sub in XRESTORMER_OPTS:
                toolbox[i] = XRestormer(subtask=sub, cache=xrestormer_cache)
                n += 1
            elif name == 'mprnet' and sub in ('denoising', 'motion_deblurring', 'deraining'):
                toolbox[i] = MPRNet(subtask=sub, cache=mprnet_cache)
                n += 1
            elif name.startswith('swinir_') and sub in SWINIR_OPTS and name[7:] in SWINIR_OPTS[sub][1]:
                toolbox[i] = SwinIR(subtask=sub, pretrained_on=name[7:], cache=swinir_cache)
                n += 1
            elif sub == 'brightening' and name in ('constant_shift', 'gamma_correction', 'histogram_equalization'):
                toolbox[i] = {'constant_shift': ConstantShift, 'gamma_correction': GammaCorrection,
                              'histogram_equalization': HistogramEqualization}[name](cache.device if cache is not None else _default_device())
                n += 1
    return n


executor = build_executor()     # singleton, like executor/__init__.py:160-169 (backends are created lazily)
