"""Cross-attention maps of the diffusion UNet, as the refinement stage consumes them.

Two parts:

* `token_attention_maps` -- pure tensor math, pinned to the reference: average the stored cross-attention probabilities
  of every 16x16 layer (all heads, both halves of the classifier-free-guidance batch: `aggregate_attention`,
  thre3d_atom/thre3d_reprs/cross_attn.py:425-436), drop the begin / end tokens, and for every requested token smooth
  (reflect pad + 3x3 Gaussian of thre3d_atom/thre3d_reprs/gaussian_smoothing.py, whose exponent is ((x - mean) / (2 std))^2),
  up-sample bilinearly to the image size and smooth again (`compute_max_attention_per_index`, :439-467).
* `CrossAttentionRecorder` -- diffusers glue (an attention processor that records the probabilities of the
  cross-attention layers up to 32x32 pixels, like `AttentionStore.forward` :174-178).  It follows the public
  `AttnProcessor` call protocol of diffusers; diffusers is not installed in the build image, so this half is untested
  here -- the refinement trainer accepts any object with `get_attn_map` / `get_num_tokens` instead.
"""
import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


def gaussian_kernel_2d(kernel_size: int = 3, sigma: float = 0.5, dtype=torch.float32) -> Tensor:
    """normalised product of 1-D kernels 1/(sigma sqrt(2 pi)) exp(-((x - mean) / (2 sigma))^2)  (gaussian_smoothing.py:30-43)"""
    ax = torch.arange(kernel_size, dtype=torch.float32)
    mean = (kernel_size - 1) / 2
    k1 = 1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(((ax - mean) / (2 * sigma)) ** 2))
    k2 = k1[:, None] * k1[None, :]
    return (k2 / k2.sum()).to(dtype)


def _smooth(image: Tensor, kernel: Tensor) -> Tensor:
    padded = F.pad(image[None, None], (1, 1, 1, 1), mode="reflect")
    return F.conv2d(padded, kernel[None, None].to(image.dtype).to(image.device))[0, 0]


def average_attention(layers: Sequence[Tensor], res: int = 16) -> Tensor:
    """layers: cross-attention probabilities [batch * heads, pixels, tokens] of the recorded layers; keeps those with
    res*res pixels and averages over layers, heads and the batch -> [res, res, tokens]"""
    picked = [a.reshape(-1, res, res, a.shape[-1]) for a in layers if a.shape[1] == res * res]
    if not picked:
        raise ValueError(f"no recorded cross-attention layer has {res}x{res} pixels")
    stacked = torch.cat(picked, dim=0)
    return stacked.sum(0) / stacked.shape[0]


def token_attention_maps(attention: Tensor, indices: Sequence[int], height: int, width: int, smooth: bool = True,
                         sigma: float = 0.5, kernel_size: int = 3) -> List[Tensor]:
    """attention [res, res, tokens] -> one [height, width] map per 1-based token index (index 1 = first prompt word)"""
    text = attention[:, :, 1:-1]
    kernel = gaussian_kernel_2d(kernel_size, sigma, attention.dtype)
    out = []
    for i in indices:
        image = text[:, :, i - 1]
        if smooth:
            image = _smooth(image, kernel)
        up = F.interpolate(image[None, None], size=(height, width), mode="bilinear", align_corners=False)[0, 0]
        if smooth:
            up = _smooth(up, kernel)
        out.append(up)
    return out


class CrossAttentionRecorder:
    """Attention processor for diffusers' `Attention` modules: computes the attention explicitly (so that the
    probabilities exist) and records those of cross-attention layers with at most `max_pixels` query positions."""

    def __init__(self, max_pixels: int = 32 ** 2):
        self.max_pixels = max_pixels
        self.records: List[Tensor] = []

    def reset(self) -> None:
        self.records = []

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kwargs):
        residual = hidden_states
        if getattr(attn, "spatial_norm", None) is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        batch, seq_len, _ = hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, seq_len, batch)
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        is_cross = encoder_hidden_states is not None
        context = hidden_states if not is_cross else encoder_hidden_states
        if is_cross and getattr(attn, "norm_cross", None):
            context = attn.norm_encoder_hidden_states(context)
        key, value = attn.to_k(context), attn.to_v(context)
        query, key, value = (attn.head_to_batch_dim(t) for t in (query, key, value))
        probs = attn.get_attention_scores(query, key, attention_mask)
        if is_cross and probs.shape[1] <= self.max_pixels:
            self.records.append(probs.detach())
        hidden_states = attn.batch_to_head_dim(torch.bmm(probs, value))
        hidden_states = attn.to_out[1](attn.to_out[0](hidden_states))
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
        if getattr(attn, "residual_connection", False):
            hidden_states = hidden_states + residual
        return hidden_states / getattr(attn, "rescale_output_factor", 1.0)


def install_recorder(unet, recorder: CrossAttentionRecorder) -> Dict[str, object]:
    """put `recorder` on every attention module of the UNet; returns the previous processors (restore with
    `unet.set_attn_processor(previous)`)"""
    previous = dict(unet.attn_processors)
    unet.set_attn_processor({name: recorder for name in previous})
    return previous
