"""PSNR helper (reference: thre3d_atom/utils/metric_utils.py:10-21)."""
import math
from typing import Any

import torch

from thre3d_atom.utils.constants import INFINITY


def mse2psnr(x: Any) -> Any:
    """-10 log10(mse); an exact zero maps to the reference's INFINITY sentinel (tensor) / inf (float)."""
    if isinstance(x, torch.Tensor):
        if float(x) == 0.0:
            return torch.full((1,), INFINITY, dtype=x.dtype, device=x.device)
        return -10.0 * torch.log10(x)
    return math.inf if x == 0.0 else -10.0 * math.log10(x)
