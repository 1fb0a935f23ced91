"""Service-class API of the reference's per-model Flask services, on the HIP backend.

Mirrors AgentApp/model_service/denoising/Restormer/model_serving.py:55-209 (class RestormerService; the
motion_deblurring / defocus_deblurring / deraining copies differ only in TASK and the YAML key) and the
``restore(image, task=...)`` convenience named by BASELINE.json.  PNG streams are decoded by the library (csrc/ira_png.cpp) and encoded
by tools.encode_png; the other containers (.jpg, .bmp) go through PIL (cv2 is not installed; PNG is lossless, so the codec does not
change the pixels).
"""
from __future__ import annotations

import io
import os
from typing import Optional

import numpy as np

from .config import SUPPORTED_TASKS, TASK_WEIGHT_FILE, canonical_subtask, task_config, SUBTASK_TO_TASK
from .tools import BackendCache, default_cache

ALLOWED_EXTENSIONS = {'png', 'jpg', 'jpeg', 'bmp'}        # model_serving.py:27


def decode_image(img_bytes, is_gray=False) -> np.ndarray:
    """model_serving.py:126-134: uint8 RGB (h,w,3), or (h,w,1) for the gray task.  PNG by the library's decoder (csrc/ira_png.cpp), the
    other containers of ALLOWED_EXTENSIONS by PIL: tools.decode_image."""
    from .tools import decode_image as _dec
    return _dec(bytes(img_bytes), gray=is_gray)


class RestormerService:
    """model_serving.py:55-209.  ``TASK`` is the service's default task (model_serving.py:22); models are
    created lazily and cached per task (``self.models``, :57,101-105)."""

    def __init__(self, task: str = 'Real_Denoising', weight_dir: Optional[str] = None, device: int = 0,
                 cache: Optional[BackendCache] = None, load: bool = False):
        if task not in SUPPORTED_TASKS:
            raise ValueError(f"Not supported task {task}")
        self.TASK = task
        self.weight_dir = weight_dir or os.environ.get('IRA_RESTORMER_WEIGHTS', '')
        if cache is None:
            dflt = default_cache()
            # the process-wide cache only when it lives on the requested device; a service asked for another GPU (or given a weight
            # directory) gets a cache of its own there
            cache = dflt if (not weight_dir and dflt.device == device) else BackendCache(device=device, weight_dir=self.weight_dir or None)
        self._cache = cache
        self.models = self._cache.models
        self.device = f'hip:{self._cache.device}'
        self.img_multiple_of = 8
        self.model = self.load_model(task) if load else None

    def get_weights_and_parameters(self, task):
        """model_serving.py:64-98: (weights path, constructor parameters)."""
        if task not in SUPPORTED_TASKS:
            raise ValueError(f"Not supported task {task}")
        return os.path.join(self.weight_dir, TASK_WEIGHT_FILE[task]), task_config(task).to_dict()

    def load_model(self, task):
        """model_serving.py:101-124."""
        return self._cache.get(task)

    def load_img_from_bytes(self, img_bytes, is_gray=False):
        """model_serving.py:126-134."""
        return decode_image(img_bytes, is_gray)

    def img_to_bytes(self, img, is_gray=False):
        """model_serving.py:146-156: PNG bytes (tools.encode_png: filter 0 rows, striped deflate - lossless like cv2.imencode, and
        0.15 s instead of PIL's 1.4 s for a 4K frame)."""
        from .tools import encode_png
        return encode_png(np.asarray(img))

    def restore_image(self, img_bytes, task, tile=None, tile_overlap=32):
        """model_serving.py:158-209: encoded image bytes in -> PNG bytes out."""
        is_gray = (task == 'Gaussian_Gray_Denoising')
        img = self.load_img_from_bytes(img_bytes, is_gray)
        be = self.load_model(task)
        restored = be.restore(img, tile=tile, tile_overlap=tile_overlap)
        return self.img_to_bytes(restored, is_gray)


def allowed_file(filename):
    """model_serving.py:216-217."""
    return '.' in filename and filename.rsplit('.', 1)[1].lower() in ALLOWED_EXTENSIONS


def restore(image, task: str = 'Real_Denoising', tool: str = 'restormer', tile=None, tile_overlap: int = 32,
            cache: Optional[BackendCache] = None):
    """restore(image, task=...) of BASELINE.json: uint8 (h,w,c) ndarray (or PIL image / path / encoded bytes)
    -> restored uint8 ndarray.  ``task`` is a reference task name (demo.py:31-36) or an agent subtask
    ('denoising', 'motion deblurring', ...: multitask_tools.py:151-156)."""
    if tool.startswith('swinir_'):
        # the agent's SwinIR tools (multitask_tools.py:81-133): task = the subtask, tool = 'swinir_15' | '_50' | '_gan' | '_psnr' | '_40'
        from .tools import SWINIR_OPTS, default_swinir_cache, read_image
        sub = canonical_subtask(task)
        if sub not in SWINIR_OPTS or tool[7:] not in SWINIR_OPTS[sub][1]:
            raise ValueError(f"tool {tool!r} does not handle task {task!r}")
        opt_task, names = SWINIR_OPTS[sub]
        if isinstance(image, (bytes, bytearray)):
            img = decode_image(bytes(image), False)
        elif isinstance(image, (str, os.PathLike)):
            img = read_image(image, False)
        elif isinstance(image, np.ndarray):
            img = image
        else:
            img = np.asarray(image.convert('RGB'), dtype=np.uint8)
        be = (cache or default_swinir_cache()).get(opt_task, names[tool[7:]])
        return be.restore(np.ascontiguousarray(img, dtype=np.uint8), tile=tile, tile_overlap=tile_overlap)
    if tool != 'restormer':
        raise ValueError(f"tool {tool!r} is not provided by the HIP backend ('restormer', 'swinir_15|50|gan|psnr|40')")
    if task not in SUPPORTED_TASKS:
        key = canonical_subtask(task)
        if key not in SUBTASK_TO_TASK:
            raise ValueError(f"Not supported task {task}")
        task = SUBTASK_TO_TASK[key]
    gray = task == 'Gaussian_Gray_Denoising'
    if isinstance(image, (bytes, bytearray)):
        img = decode_image(bytes(image), gray)
    elif isinstance(image, (str, os.PathLike)):
        from .tools import read_image
        img = read_image(image, gray)
    elif isinstance(image, np.ndarray):
        img = image if image.ndim == 3 else image[:, :, None]
    else:   # PIL image
        img = np.asarray(image.convert('L' if gray else 'RGB'), dtype=np.uint8)
        img = img[:, :, None] if img.ndim == 2 else img
    be = (cache or default_cache()).get(task)
    return be.restore(np.ascontiguousarray(img, dtype=np.uint8), tile=tile, tile_overlap=tile_overlap)


def make_app(service: Optional[RestormerService] = None):
    """Flask app with the reference's routes (/health :220, /tasks :230, /restore :245-320), form-data and
    JSON/base64 request forms.  Imported lazily: flask is optional."""
    import base64
    from flask import Flask, jsonify, request, send_file
    svc = service or RestormerService()
    app = Flask(__name__)
    app.config['MAX_CONTENT_LENGTH'] = 16 * 1024 * 1024     # model_serving.py:28-29

    @app.route('/health', methods=['GET'])
    def health():
        return jsonify({'status': 'healthy', 'service': 'restormer-inference', 'device': svc.device,
                        'loaded_models': list(svc.models.keys())}), 200

    @app.route('/tasks', methods=['GET'])
    def tasks():
        return jsonify({'tasks': SUPPORTED_TASKS, 'descriptions': {t: '' for t in SUPPORTED_TASKS}}), 200

    @app.route('/restore', methods=['POST'])
    def restore_route():
        try:
            task = svc.TASK
            if request.is_json:
                data = request.get_json()
                b64 = data.get('image_base64')
                if not b64:
                    return jsonify({'error': 'Lack of base64 parameters'}), 400
                try:
                    img_bytes = base64.b64decode(b64)
                except Exception as e:   # noqa: BLE001
                    return jsonify({'error': f'Failed base64 decoding: {e}'}), 400
                tile, ov = data.get('tile'), data.get('tile_overlap', 32)
            else:
                if 'image' not in request.files:
                    return jsonify({'error': 'Lack of image file'}), 400
                f = request.files['image']
                if f.filename == '':
                    return jsonify({'error': 'No file'}), 400
                if not allowed_file(f.filename):
                    return jsonify({'error': 'unsupported file type', 'allowed_types': sorted(ALLOWED_EXTENSIONS)}), 400
                img_bytes = f.read()
                tile, ov = request.form.get('tile'), request.form.get('tile_overlap', 32)
            tile = int(tile) if tile else None
            out = svc.restore_image(img_bytes, task, tile, int(ov))
            return send_file(io.BytesIO(out), mimetype='image/png', as_attachment=True, download_name='restored.png')
        except ValueError as e:
            return jsonify({'error': str(e)}), 400
        except Exception as e:   # noqa: BLE001
            return jsonify({'error': 'Failed image restoration', 'message': str(e)}), 500

    return app
