"""Tiny deterministic stand-ins for the Stable-Diffusion stack (diffusers / transformers objects) -- test infrastructure.

Neither `diffusers` nor SD weights exist in this image (no network), so the SDS boundary
(thre3d_atom/thre3d_reprs/sd.py: StableDiffusion.train_step :174-234, SpecifyGradient :20-34,
scoreDistillationLoss.training_step :365-385) is pinned with stand-ins that have the same CALL SURFACE as the objects
the reference uses -- `AutoencoderKL.encode(x).latent_dist.sample()`, `UNet2DConditionModel(x, t,
encoder_hidden_states=...).sample`, `DDIMScheduler.{config.num_train_timesteps, set_timesteps, alphas_cumprod,
add_noise}`, `CLIPTokenizer(...)`, `CLIPTextModel(ids)[0]` -- and small fixed-seed weights.  What gets pinned is
everything AROUND the networks: image layout / 512^2 bilinear resize / 2x-1 / 0.18215 scaling, timestep range and
schedule, noise injection, classifier-free guidance (scale 100), w(t) = 1 - alpha_bar_t, nan_to_num, the gradient
injected at the latents and its path back through the (differentiated) VAE encoder to the rendered colours.

`tools/gen_golden.py` runs the REFERENCE's sd.py on these stand-ins and records inputs, random draws and outputs into
tests/golden/sds_boundary.npz; tests/test_sds_boundary.py runs the build's sd.py on the same stand-ins.
`installed()` swaps fake `diffusers` / `transformers` modules into sys.modules for the duration of a `with` block."""
import contextlib
import math
import sys
import types

import torch
from torch import nn

EMBED_DIM = 16
VOCAB = 49408
END_TOKEN = 49407          # CLIP's end-of-text / padding id (sd.py:110)
BEGIN_TOKEN = 49406


def _seeded(module: nn.Module, seed: int) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / math.sqrt(max(p[0].numel(), 1))))
    return module.requires_grad_(True)


class _FromPretrained:
    @classmethod
    def from_pretrained(cls, model_key, subfolder=None, use_auth_token=None, **kwargs):
        obj = cls()
        obj.loaded_from = (model_key, subfolder, use_auth_token)
        return obj


class _LatentDist:
    """diffusers' DiagonalGaussianDistribution: sample() = mean + std * randn_like(mean)"""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))

    def sample(self):
        return self.mean + self.std * torch.randn_like(self.mean)


class _PatchLinear(nn.Module):
    """a convolution whose kernel equals its stride, written as patchify + matmul: no MIOpen involved (its backward-data
    kernel for these shapes faults on this ROCm stack -- 'Memory access fault' -- depending on the allocator state;
    found with the full-size SDS loop test, reproduced with torch.backends.cudnn.enabled = False as the cure)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.k, self.lin = k, nn.Linear(cin * k * k, cout)

    def forward(self, x):
        b, c, h, w = x.shape
        k = self.k
        p = x.reshape(b, c, h // k, k, w // k, k).permute(0, 2, 4, 1, 3, 5).reshape(b, h // k, w // k, c * k * k)
        return self.lin(p).permute(0, 3, 1, 2)


class AutoencoderKL(nn.Module, _FromPretrained):
    def __init__(self):
        super().__init__()
        self.enc = nn.Sequential(_PatchLinear(3, 8, 4), nn.SiLU(), _PatchLinear(8, 8, 2))   # /8 like the SD VAE
        _seeded(self, 101)

    def encode(self, imgs):
        return types.SimpleNamespace(latent_dist=_LatentDist(self.enc(imgs)))


class UNet2DConditionModel(nn.Module, _FromPretrained):
    in_channels = 4

    def __init__(self):
        super().__init__()
        self.lin_in = nn.Linear(4, 8)          # pointwise (1x1) layers + a fixed 3x3 box filter for spatial mixing
        self.text = nn.Linear(EMBED_DIM, 8)
        self.lin_out = nn.Linear(8, 4)
        _seeded(self, 202)

    def forward(self, x, t, encoder_hidden_states=None):
        temb = torch.sin(t.to(x.dtype).reshape(-1, 1, 1, 1) * 0.01)
        ctx = self.text(encoder_hidden_states).mean(dim=1)[:, :, None, None]   # [B, 8, 1, 1]: text-dependent shift
        mixed = x + torch.nn.functional.avg_pool2d(x, 3, 1, 1)
        h = torch.tanh(self.lin_in(mixed.permute(0, 2, 3, 1)).permute(0, 3, 1, 2) + ctx + temb)
        return types.SimpleNamespace(sample=self.lin_out(h.permute(0, 2, 3, 1)).permute(0, 3, 1, 2))


class DDIMScheduler(_FromPretrained):
    """the SD schedule: scaled-linear betas 0.00085 .. 0.012 over 1000 steps; add_noise as in diffusers"""

    def __init__(self):
        self.config = types.SimpleNamespace(num_train_timesteps=1000)
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.timesteps = torch.arange(num_inference_steps - 1, -1, -1, device=device)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod.to(original_samples.device)[timesteps]
        sa, sb = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class PNDMScheduler(DDIMScheduler):
    pass


class _Encoding(dict):
    @property
    def input_ids(self):
        return self["input_ids"]


class CLIPTokenizer(_FromPretrained):
    model_max_length = 77

    def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
        texts = [text] if isinstance(text, str) else list(text)
        n = max_length or self.model_max_length
        rows = []
        for s in texts:
            words = s.replace(",", " , ").split()
            ids = [BEGIN_TOKEN] + [1 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 40000) for w in words][: n - 2]
            rows.append(ids + [END_TOKEN] * (n - len(ids)))
        return _Encoding(input_ids=torch.tensor(rows, dtype=torch.long))


class CLIPTextModel(nn.Module, _FromPretrained):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(VOCAB, EMBED_DIM)
        self.pos = nn.Parameter(torch.zeros(77, EMBED_DIM))
        _seeded(self, 303)

    def forward(self, input_ids):
        return (self.emb(input_ids) + self.pos[: input_ids.shape[1]],)


@contextlib.contextmanager
def installed():
    """fake `diffusers` and `transformers` (+ an empty `cv2`, imported by the reference's cross_attn.py) in sys.modules"""
    fake_d = types.ModuleType("diffusers")
    for cls in (AutoencoderKL, UNet2DConditionModel, DDIMScheduler, PNDMScheduler):
        setattr(fake_d, cls.__name__, cls)
    fake_t = types.ModuleType("transformers")
    fake_t.CLIPTokenizer, fake_t.CLIPTextModel = CLIPTokenizer, CLIPTextModel
    fake_t.logging = types.SimpleNamespace(set_verbosity_error=lambda: None)
    saved = {k: sys.modules.get(k) for k in ("diffusers", "transformers", "cv2")}
    sys.modules["diffusers"], sys.modules["transformers"] = fake_d, fake_t
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


class RandomTape:
    """record (`record=True`) or replay the torch.randint / torch.randn_like draws of a guidance step, so that the
    build's sd.py can be fed the very timesteps and noises the reference drew (CPU and GPU generators differ)"""

    def __init__(self, record=True, draws=None, device=None):
        self.record, self.draws, self.device = record, list(draws or []), device
        self._cursor = 0
        self._orig = (torch.randint, torch.randn_like)

    def __enter__(self):
        orig_randint, orig_randn_like = self._orig

        def randint(*args, **kwargs):
            if self.record:
                out = orig_randint(*args, **kwargs)
                self.draws.append(out.detach().cpu().clone())
                return out
            out = self.draws[self._cursor].to(kwargs.get("device") or self.device)
            self._cursor += 1
            return out

        def randn_like(x, **kwargs):
            if self.record:
                out = orig_randn_like(x, **kwargs)
                self.draws.append(out.detach().cpu().clone())
                return out
            out = self.draws[self._cursor].to(device=x.device, dtype=x.dtype)
            self._cursor += 1
            assert out.shape == x.shape
            return out

        torch.randint, torch.randn_like = randint, randn_like
        return self

    def __exit__(self, *exc):
        torch.randint, torch.randn_like = self._orig
        return False
