"""Restormer configuration table.

Mirrors the reference's task -> (architecture parameters, LayerNorm type, weight file) table:
  AgenticIR/executor/defocus_deblurring/tools/Restormer/demo.py:52-70,92
  AgentApp/model_service/denoising/Restormer/model_serving.py:64-98
and the agent subtask -> task mapping of AgenticIR/executor/multitask_tools.py:151-157.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field, asdict

# demo.py:31-36 / model_serving.py:32-39
SUPPORTED_TASKS = [
    'Motion_Deblurring',
    'Single_Image_Defocus_Deblurring',
    'Deraining',
    'Real_Denoising',
    'Gaussian_Gray_Denoising',
    'Gaussian_Color_Denoising',
]

# multitask_tools.py:151-157 (keys with '_' as the toolboxes pass them, and with ' ' / '-' as
# executor.toolbox_router and model_services.yaml spell them: SURVEY App. B item 9)
SUBTASK_TO_TASK = {
    'denoising': 'Real_Denoising',
    'motion_deblurring': 'Motion_Deblurring',
    'defocus_deblurring': 'Single_Image_Defocus_Deblurring',
    'deraining': 'Deraining',
}

# weight file basenames, demo.py:53-69
TASK_WEIGHT_FILE = {
    'Motion_Deblurring': 'motion_deblurring.pth',
    'Single_Image_Defocus_Deblurring': 'single_image_defocus_deblurring.pth',
    'Deraining': 'deraining.pth',
    'Real_Denoising': 'real_denoising.pth',
    'Gaussian_Color_Denoising': 'gaussian_color_denoising_blind.pth',
    'Gaussian_Gray_Denoising': 'gaussian_gray_denoising_blind.pth',
}


def canonical_subtask(name: str) -> str:
    """'motion deblurring' / 'motion_deblurring' / 'super-resolution' -> underscore form."""
    return name.strip().lower().replace(' ', '_').replace('-', '_')


@dataclass
class RestormerConfig:
    """Constructor arguments of the reference ``Restormer`` (restormer_arch.py:193-205)."""
    inp_channels: int = 3
    out_channels: int = 3
    dim: int = 48
    num_blocks: list = field(default_factory=lambda: [4, 6, 6, 8])
    num_refinement_blocks: int = 4
    heads: list = field(default_factory=lambda: [1, 2, 4, 8])
    ffn_expansion_factor: float = 2.66
    bias: bool = False
    LayerNorm_type: str = 'WithBias'
    dual_pixel_task: bool = False

    def to_dict(self) -> dict:
        return copy.deepcopy(asdict(self))

    def hidden(self, dim: int) -> int:
        # restormer_arch.py:80  hidden_features = int(dim*ffn_expansion_factor)
        return int(dim * self.ffn_expansion_factor)


def task_config(task: str) -> RestormerConfig:
    """demo.py:52-70 (get_weights_and_parameters) without the weight path."""
    if task not in SUPPORTED_TASKS:
        raise ValueError(f"Not supported task {task}")   # model_serving.py:95-96
    cfg = RestormerConfig()
    if task in ('Real_Denoising', 'Gaussian_Color_Denoising', 'Gaussian_Gray_Denoising'):
        cfg.LayerNorm_type = 'BiasFree'
    if task == 'Gaussian_Gray_Denoising':
        cfg.inp_channels = 1
        cfg.out_channels = 1
    return cfg


def reduced_config(LayerNorm_type: str = 'BiasFree', inp_channels: int = 3) -> RestormerConfig:
    """Small variant used for fast parity cases (dim=16, blocks [1,1,1,1], refine 1)."""
    return RestormerConfig(inp_channels=inp_channels, out_channels=inp_channels, dim=16,
                           num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
                           heads=[1, 2, 4, 8], LayerNorm_type=LayerNorm_type)
