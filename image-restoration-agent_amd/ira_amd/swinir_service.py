"""Service-module API of the reference's SwinIR Flask service, on the HIP backend.

Mirrors AgentApp/model_service/super_resolution/SwinIR/model_serving.py: ``define_model`` (:43-103), ``load_model`` with its
cache key (:105-128), ``process_image`` (:131-175: float32 HWC **BGR** in [0,1] -> uint8 HWC BGR, always-pad flip-concat, whole
or tiled, crop, clamp, round) and the routes /health, /predict, /predict_batch (:178-334).  PNG decode is the library's (csrc/ira_png.cpp), PNG encode tools.encode_png, other containers PIL (cv2 is
not installed); PNG is lossless so the library does not change the pixels.
"""
from __future__ import annotations

import base64
import io
import os
from typing import Callable, Optional

import numpy as np

from .swinir import PARAM_KEY, SwinIRConfig, task_config


def define_model(task, scale=1, noise=15, jpeg=40, large_model=False):
    """model_serving.py:43-103: (constructor arguments, checkpoint key).  classical_sr trains on 48-pixel patches there (:45)."""
    if task == 'lightweight_sr':
        raise ValueError("lightweight_sr (pixelshuffledirect upsampler) is not built on the HIP backend")
    try:
        cfg = task_config(task, scale=scale, large_model=large_model, training_patch_size=48)
    except ValueError:
        raise ValueError(f"Unknown task: {task}")
    return cfg, PARAM_KEY.get(task, 'params')


class SwinIRService:
    """The module-level state of model_serving.py (``models`` dict, ``model_path``, ``device``) as an object."""

    def __init__(self, model_path: Optional[str] = None, device: int = 0, state_dict_loader: Optional[Callable] = None,
                 backend_factory: Optional[Callable] = None):
        self.model_path = model_path or os.environ.get('IRA_SWINIR_MODEL_PATH', '')
        self.device = device
        self.models: dict = {}
        self._loader = state_dict_loader
        self._factory = backend_factory

    def _state_dict(self, key):
        if self._loader is not None:
            return self._loader(self.model_path, key)
        import torch
        ckpt = torch.load(self.model_path, map_location='cpu')            # :114-118
        return ckpt[key] if key in ckpt else ckpt

    def load_model(self, task, scale=1, noise=15, jpeg=40, large_model=False):
        """:105-128: (model, None) or (None, error text); cached under the reference's key."""
        model_key = f"{task}_{scale}_{noise}_{jpeg}_{large_model}"
        if model_key not in self.models:
            if self._loader is None and not os.path.exists(self.model_path):
                return None, f"Model file not found: {self.model_path}"
            try:
                cfg, key = define_model(task, scale, noise, jpeg, large_model)
                sd = self._state_dict(key)
                if self._factory is not None:
                    self.models[model_key] = self._factory(cfg, sd, self.device)
                else:
                    from .backend import SwinIRBackend
                    self.models[model_key] = SwinIRBackend(cfg, sd, device=self.device)
            except Exception as e:   # noqa: BLE001  (the reference reports every failure as text, :124-126)
                return None, f"Error loading model: {e}"
        return self.models[model_key], None

    @staticmethod
    def process_image(img_lq, model, scale, window_size, tile=None, tile_overlap=32):
        """:131-175.  ``img_lq``: float32 (h,w,c) BGR in [0,1] as ``cv2.imdecode(...)/255`` gives it (:228-232); returns uint8 BGR.
        The backend harness works on uint8 RGB: x/255*255 rounds back to x exactly, and the BGR<->RGB flips (:133-136,171-172)
        are applied around it."""
        a = np.asarray(img_lq, dtype=np.float32)
        u8 = np.rint(a * 255.0)
        if np.abs(u8 - a * 255.0).max() > 1e-3:
            raise ValueError("process_image expects an image decoded from 8-bit data (values k/255)")
        u8 = u8.astype(np.uint8)
        rgb = u8 if u8.shape[2] == 1 else u8[:, :, ::-1]
        cfg: SwinIRConfig = model.cfg
        if cfg.upscale != scale or cfg.window_size != window_size:
            raise ValueError(f"model was built for scale {cfg.upscale} / window {cfg.window_size}, got {scale} / {window_size}")
        out = model.restore(np.ascontiguousarray(rgb), tile=tile, tile_overlap=tile_overlap)
        return out if out.shape[2] == 1 else np.ascontiguousarray(out[:, :, ::-1])


def _decode_bgr(data: bytes):
    from .tools import decode_image
    try:
        rgb = decode_image(bytes(data))
    except Exception:   # noqa: BLE001
        return None
    return rgb[:, :, ::-1].astype(np.float32) / 255.          # cv2.IMREAD_COLOR order, :228-232


def _encode_png_bgr(bgr: np.ndarray) -> bytes:
    from .tools import encode_png
    return encode_png(np.ascontiguousarray(bgr[:, :, ::-1]))


def make_app(service: Optional[SwinIRService] = None):
    """Flask app with the reference's routes and form fields (:178-334)."""
    from flask import Flask, jsonify, request, send_file
    svc = service or SwinIRService()
    app = Flask(__name__)

    def _args():
        f = request.form
        tile = f.get('tile', None)
        return (f.get('task', 'real_sr'), int(f.get('scale', 1)), int(f.get('noise', 15)), int(f.get('jpeg', 40)),
                f.get('large_model', 'false').lower() == 'true', int(tile) if tile else None, int(f.get('tile_overlap', 32)))

    @app.route('/health', methods=['GET'])
    def health_check():
        return jsonify({'status': 'ok', 'device': f'hip:{svc.device}', 'loaded_models': len(svc.models)})

    @app.route('/predict', methods=['POST'])
    def predict():
        try:
            if 'image' not in request.files:
                return jsonify({'error': 'No image file provided'}), 400
            file = request.files['image']
            if file.filename == '':
                return jsonify({'error': 'Empty filename'}), 400
            task, scale, noise, jpeg, large, tile, ov = _args()
            if task not in ['real_sr', 'color_dn', 'color_jpeg_car']:
                return jsonify({'error': 'Invalid task type'}), 400
            ws = 7 if task in ['jpeg_car', 'color_jpeg_car'] else 8
            model, error = svc.load_model(task, scale, noise, jpeg, large)
            if error:
                return jsonify({'error': error}), 500
            img = _decode_bgr(file.read())
            if img is None:
                return jsonify({'error': 'Failed to decode image'}), 400
            png = _encode_png_bgr(svc.process_image(img, model, scale, ws, tile, ov))
            if request.form.get('return_type', 'file') == 'base64':
                return jsonify({'success': True, 'image': base64.b64encode(png).decode('utf-8'), 'format': 'png'})
            return send_file(io.BytesIO(png), mimetype='image/png')
        except Exception as e:   # noqa: BLE001
            return jsonify({'error': str(e)}), 500

    @app.route('/predict_batch', methods=['POST'])
    def predict_batch():
        try:
            if 'images' not in request.files:
                return jsonify({'error': 'No image files provided'}), 400
            files = request.files.getlist('images')
            if not files:
                return jsonify({'error': 'Empty file list'}), 400
            task, scale, noise, jpeg, large, tile, ov = _args()
            ws = 7 if task in ['jpeg_car', 'color_jpeg_car'] else 8
            model, error = svc.load_model(task, scale, noise, jpeg, large)
            if error:
                return jsonify({'error': error}), 500
            results = []
            for file in files:
                try:
                    img = _decode_bgr(file.read())
                    if img is None:
                        results.append({'filename': file.filename, 'error': 'Failed to decode image'})
                        continue
                    png = _encode_png_bgr(svc.process_image(img, model, scale, ws, tile, ov))
                    results.append({'filename': file.filename, 'success': True, 'image': base64.b64encode(png).decode('utf-8')})
                except Exception as e:   # noqa: BLE001
                    results.append({'filename': file.filename, 'error': str(e)})
            return jsonify({'results': results})
        except Exception as e:   # noqa: BLE001
            return jsonify({'error': str(e)}), 500

    return app
