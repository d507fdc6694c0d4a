"""Occupancy model of the accumulator stage.

`density2occupancy_pb` is a *value* stored in SHVoxGridRenderConfig and pickled by qualified name
into checkpoints (reference: thre3d_atom/rendering/volumetric/accumulate.py:24-28), so it must be
importable from here.  The accumulation itself (density -> alpha -> transmittance -> compositing,
accumulate.py:31-198) runs inside the fused HIP kernels; the renderer checks that the configured
callable is this one and refuses anything else.
"""
import torch
from torch import Tensor


def density2occupancy_pb(densities: Tensor, deltas: Tensor) -> Tensor:
    """alpha = 1 - exp(-sigma * delta) (Beer-Lambert)."""
    return 1.0 - torch.exp(-(densities * deltas))
