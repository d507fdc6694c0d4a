"""Held-out evaluation of a trained model (interface of the reference's thre3d_atom/modules/testers.py:17-71).

Every test view is one fused HIP forward launch (`optimized_sampling=True`, `render_num_samples_per_ray` samples, like
the reference); PSNR is computed on the GPU.  LPIPS needs the external `lpips` network: it is reported when that
package is importable and skipped otherwise (the reference hard-requires it)."""
from typing import Any, Dict, Optional

import numpy as np
import torch
from torch.nn.functional import mse_loss

from thre3d_atom.modules.volumetric_model import VolumetricModel
from thre3d_atom.utils.imaging_utils import CameraPose
from thre3d_atom.utils.logging import log
from thre3d_atom.utils.metric_utils import mse2psnr


def test_sh_vox_grid_vol_mod_with_posed_images(
    vol_mod: VolumetricModel,
    test_dl: Any,                       # DataLoader (batch size 1) or a posed-images dataset
    parallel_rays_chunk_size: Optional[int] = None,
    tensorboard_writer: Any = None,
    global_step: Optional[int] = None,
) -> Dict[str, float]:
    dataset = getattr(test_dl, "dataset", test_dl)
    intrinsics = dataset.camera_intrinsics
    log.info(f"Testing the model on {len(dataset)} heldout images")
    lpips_net = None
    try:
        import lpips  # noqa: WPS433 (optional dependency)

        lpips_net = lpips.LPIPS(net="vgg").to(vol_mod.device)
    except ImportError:
        pass
    psnrs, lpipss = [], []
    for index in range(len(dataset)):
        image, pose, _ = dataset[index]
        image, pose = image.to(vol_mod.device), pose
        rendered = vol_mod.render(
            camera_pose=CameraPose(rotation=pose[:, :3], translation=pose[:, 3:]), camera_intrinsics=intrinsics,
            parallel_rays_chunk_size=parallel_rays_chunk_size, gpu_render=True, optimized_sampling=True,
            num_samples_per_ray=vol_mod.render_config.render_num_samples_per_ray)
        colour = rendered.colour.permute(2, 0, 1)
        with torch.no_grad():
            psnrs.append(float(mse2psnr(mse_loss(colour, image).item())))
            if lpips_net is not None:
                lpipss.append(float(lpips_net(colour[None], image[None], normalize=True).item()))
    out = {"psnr": float(np.mean(psnrs))}
    log.info(f"Mean PSNR on holdout set: {out['psnr']}")
    if lpipss:
        out["lpips"] = float(np.mean(lpipss))
        log.info(f"Mean LPIPS on holdout set: {out['lpips']}")
    if tensorboard_writer is not None and global_step is not None:
        tensorboard_writer.add_scalar("TEST_SET_PSNR", out["psnr"], global_step=global_step)
        if "lpips" in out:
            tensorboard_writer.add_scalar("TEST_SET_LPIPS", out["lpips"], global_step=global_step)
    return out


test_sh_vox_grid_vol_mod_with_posed_images.__test__ = False  # a library function whose reference name starts with "test_"
