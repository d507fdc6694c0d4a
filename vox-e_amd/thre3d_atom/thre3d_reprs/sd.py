"""Score-distillation guidance boundary.

The Stable-Diffusion VAE/UNet/CLIP stack stays under PyTorch-ROCm (MFMA work, out of scope as kernels --
BASELINE.json north_star); what matters to the render hot path is the tensor boundary: rendered colour
[H*W, 3] in, dL/dcolour out.  This module keeps the reference's class names and call signatures
(thre3d_atom/thre3d_reprs/sd.py: `SpecifyGradient` :20-34, `StableDiffusion` :43-330,
`scoreDistillationLoss` :333-384) on top of `diffusers`; the heavy imports are lazy so the renderer does not
depend on them.  Any object with `training_step(colour, H, W, directions=..., global_step=...)` and
`get_current_max_step_ratio()` can be passed to the SDS trainer as guidance instead.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

_MODEL_KEYS = {
    "2.1": "stabilityai/stable-diffusion-2-1-base",
    "2.0": "stabilityai/stable-diffusion-2-base",
    "1.5": "runwayml/stable-diffusion-v1-5",
    "1.4": "CompVis/stable-diffusion-v1-4",
}


class SpecifyGradient(torch.autograd.Function):
    """loss stand-in whose backward injects a precomputed gradient (divided by the batch size)"""

    @staticmethod
    def forward(ctx, input_tensor, gt_grad):
        ctx.save_for_backward(gt_grad)
        return torch.zeros([1], device=input_tensor.device, dtype=input_tensor.dtype)

    @staticmethod
    def backward(ctx, grad):
        (gt_grad,) = ctx.saved_tensors
        return gt_grad / len(gt_grad), None


def seed_everything(seed: int) -> None:
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


class StableDiffusion(nn.Module):
    """Frozen SD pipeline pieces + the SDS `train_step` (noise-residual gradient at the latents)."""

    def __init__(self, device, sd_version: str = "2.1", hf_key: Optional[str] = None, t_sched_start: int = 1500,
                 t_sched_freq: int = 500, t_sched_gamma: float = 1.0, auth_token=None):
        super().__init__()
        try:
            from diffusers import AutoencoderKL, DDIMScheduler, UNet2DConditionModel
            from transformers import CLIPTextModel, CLIPTokenizer
        except ImportError as exc:  # the renderer itself never needs these
            raise ImportError(
                "score distillation needs `diffusers` and `transformers` (and downloaded SD weights); "
                "pass your own guidance object to the SDS trainer if they are unavailable"
            ) from exc
        if hf_key is None and sd_version not in _MODEL_KEYS:
            raise ValueError(f"Stable-diffusion version {sd_version} not supported.")
        key = hf_key if hf_key is not None else _MODEL_KEYS[sd_version]
        token = auth_token if sd_version == "1.4" else False
        self.device = device
        self.t_sched_start, self.t_sched_freq, self.t_sched_gamma = t_sched_start, t_sched_freq, t_sched_gamma
        self.vae = AutoencoderKL.from_pretrained(key, subfolder="vae", use_auth_token=token).to(device)
        self.tokenizer = CLIPTokenizer.from_pretrained(key, subfolder="tokenizer", use_auth_token=token)
        self.text_encoder = CLIPTextModel.from_pretrained(key, subfolder="text_encoder", use_auth_token=token).to(device)
        self.unet = UNet2DConditionModel.from_pretrained(key, subfolder="unet", use_auth_token=token).to(device)
        self.scheduler = DDIMScheduler.from_pretrained(key, subfolder="scheduler")
        for module in (self.vae, self.text_encoder, self.unet):
            module.requires_grad_(False)
        self.num_train_timesteps = self.scheduler.config.num_train_timesteps
        self.min_step_ratio, self.max_step_ratio = 0.02, 0.98
        self.min_step = int(self.num_train_timesteps * self.min_step_ratio)
        self.max_step = int(self.num_train_timesteps * self.max_step_ratio)
        self.alphas = self.scheduler.alphas_cumprod.to(device)

    def get_max_step_ratio(self) -> float:
        return self.max_step_ratio

    def get_num_tokens(self, prompt: str) -> int:
        """tokens of the padded prompt that are not the end / padding token 49407 (begin token included; sd.py:104-114)"""
        ids = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt")["input_ids"][0]
        return int((ids != 49407).sum())

    @torch.no_grad()
    def get_attn_map(self, prompt: str, pred_rgb: torch.Tensor, timestamp: int = 0, indices_to_fetch=(7,),
                     guidance_scale: float = 100, logvar=None):
        """Cross-attention maps of `prompt`'s tokens for the rendered image (one noisy UNet evaluation at `timestamp`,
        or at a random step when it is 0): list of [H, W] maps for the 1-based `indices_to_fetch`, and the step
        (sd.py:138-172).  Uses an explicit attention processor while it runs and restores the UNet's own afterwards."""
        from thre3d_atom.thre3d_reprs.cross_attn import (
            CrossAttentionRecorder, average_attention, install_recorder, token_attention_maps)

        height, width = pred_rgb.shape[-2:]
        text_embeddings = self.get_text_embeds(prompt, "")
        pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
        t = torch.randint(self.min_step, self.max_step + 1, [1], dtype=torch.long, device=self.device)
        if timestamp > 0:
            t = torch.as_tensor(timestamp, dtype=torch.long, device=self.device)
        latents = self.encode_imgs(pred_rgb_512)
        noisy = self.scheduler.add_noise(latents, torch.randn_like(latents), t)
        recorder = CrossAttentionRecorder()
        previous = install_recorder(self.unet, recorder)
        try:
            self.unet(torch.cat([noisy] * 2), t, encoder_hidden_states=text_embeddings)
        finally:
            self.unet.set_attn_processor(previous)
        maps = None
        if indices_to_fetch is not None:
            maps = token_attention_maps(average_attention(recorder.records, res=16), indices_to_fetch, height, width)
        return maps, int(t.item())

    @torch.no_grad()
    def get_text_embeds(self, prompt: str, negative_prompt: str) -> torch.Tensor:
        def embed(text):
            tokens = self.tokenizer([text], padding="max_length", max_length=self.tokenizer.model_max_length,
                                    truncation=True, return_tensors="pt")
            return self.text_encoder(tokens.input_ids.to(self.device))[0]

        return torch.cat([embed(negative_prompt), embed(prompt)])

    def encode_imgs(self, imgs: torch.Tensor) -> torch.Tensor:
        posterior = self.vae.encode(2 * imgs - 1).latent_dist
        return posterior.sample() * 0.18215

    def train_step(self, text_embeddings, pred_rgb, guidance_scale: float = 100, global_step: int = -1, logvar=None):
        if global_step >= self.t_sched_start and global_step % self.t_sched_freq == 0:
            self.max_step_ratio = max(self.max_step_ratio * self.t_sched_gamma, 0.22)
        self.max_step = int(self.num_train_timesteps * self.max_step_ratio)
        pred_rgb_512 = F.interpolate(pred_rgb, (512, 512), mode="bilinear", align_corners=False)
        t = torch.randint(self.min_step, self.max_step + 1, [1], dtype=torch.long, device=self.device)
        latents = self.encode_imgs(pred_rgb_512)  # differentiated (VAE encoder), like the reference
        with torch.no_grad():
            noise = torch.randn_like(latents)
            noisy = self.scheduler.add_noise(latents, noise, t)
            pred = self.unet(torch.cat([noisy] * 2), t, encoder_hidden_states=text_embeddings).sample
            uncond, text = pred.chunk(2)
            pred = text + guidance_scale * (text - uncond)
            grad = torch.nan_to_num((1 - self.alphas[t]) * (pred - noise))
            if logvar is not None:
                grad = grad * torch.exp(-1 * logvar)
        return SpecifyGradient.apply(latents, grad)


class scoreDistillationLoss(nn.Module):  # noqa: N801 (reference class name)
    def __init__(self, device, prompt: str, t_sched_start: int = 1500, t_sched_freq: int = 500,
                 t_sched_gamma: float = 1.0, directional: bool = True, sd_version: str = "2.0"):
        super().__init__()
        self.directional = directional
        self.sd_model = StableDiffusion(device, sd_version, t_sched_start=t_sched_start, t_sched_freq=t_sched_freq,
                                        t_sched_gamma=t_sched_gamma)
        if directional:
            self.text_encodings: Dict[str, torch.Tensor] = {
                view: self.sd_model.get_text_embeds(f"{prompt}, {view} view", "")
                for view in ("side", "overhead", "back", "front")
            }
        else:
            self.text_encoding = self.sd_model.get_text_embeds(prompt, "")

    def get_current_max_step_ratio(self) -> float:
        return self.sd_model.get_max_step_ratio()

    def training_step(self, output, image_height: int, image_width: int, directions: Optional[List[str]] = None,
                      global_step: int = -1, logvars=None):
        imgs = output.reshape(-1, image_height, image_width, 3).permute(0, 3, 1, 2)
        if not self.directional:
            return self.sd_model.train_step(self.text_encoding, imgs, global_step=global_step, logvar=logvars)
        if directions is None:
            raise AssertionError("Must supply direction if SDS loss is set to directional mode")
        loss = 0
        for idx, view in enumerate(directions):
            logvar = None if logvars is None else logvars[idx]
            loss = loss + self.sd_model.train_step(self.text_encodings[view], imgs, global_step=global_step, logvar=logvar)
        return loss
