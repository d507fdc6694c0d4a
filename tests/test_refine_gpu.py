"""GPU tests of the refinement stage callers: build_graph / get_edit_region on the reference's recorded cases, the
attention-grid training loop with a stand-in attention source, the splice and the connected-component clean-up."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_refine import pooled_inputs

from oracle import voxe_oracle as vo

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from thre3d_atom.modules.attn_grid_trainer import refine_edited_relu_field, splice_reference_outside_edit_region
    from thre3d_atom.modules.refinement_functions import (
        build_graph,
        get_edit_region,
        restore_outside_largest_component,
    )
    from thre3d_atom.modules.volumetric_model import VolumetricModel, create_volumetric_model_from_saved_model_attn
    from thre3d_atom.thre3d_reprs.renderers import (
        SHVoxGridRenderConfig,
        render_sh_voxel_grid,
        render_sh_voxel_grid_attn,
    )
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelSize, create_voxel_grid_from_saved_info_dict_attn
    from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics

    DEV = torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_build_graph_on_reference_cases(tag):
    """product build_graph (HIP construction + HIP cut, torch seed selection) == oracle cut of the oracle graph with
    the t-links the REFERENCE assigned; node order as the reference's idx_values"""
    z = load_golden("refine_graph.npz")
    kw = {k[len(tag) + 4:]: z[k].item() for k in z.files if k.startswith(f"{tag}_kw_")}
    dev = lambda name: torch.from_numpy(z[f"{tag}_{name}"]).to(DEV)  # noqa: E731
    torch.manual_seed(int(z[f"{tag}_seed"]))
    segments, idx_values = build_graph(dev("features"), dev("densities"), dev("edit_attn"), dev("obj_attn"), **kw)
    idx = z[f"{tag}_node_idx"]
    assert np.array_equal(idx_values.numpy(), idx)
    dens, feat, dilate = pooled_inputs(z, tag)
    node, cap = vo.graph_build(dens[..., 0], feat, kw["sigma"], dilate)
    term = np.zeros(node.shape, np.int8)
    for i, s, t in z[f"{tag}_tedges"]:
        term[tuple(idx[int(i)])] = 1 if np.isinf(s) else -1
    seg, _, _ = vo.graphcut(node, term, cap)
    assert np.array_equal(segments.numpy(), seg[tuple(idx.T)].astype(np.int64))
    assert set(np.unique(segments.numpy())) <= {0, 1}


def _scene_models(side=32, samples=48):
    """edited field = a sphere with a "hat" (small blob on top) in a different colour; reference field = sphere only"""
    ax = (torch.arange(side, dtype=torch.float32) + 0.5) / side * 3.0 - 1.5
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    body = torch.sqrt(x * x + y * y + z * z) < 0.8
    hat = torch.sqrt(x * x + y * y + (z - 0.95) ** 2) < 0.35
    vs = VoxelSize(3.0 / side, 3.0 / side, 3.0 / side)
    cfg = SHVoxGridRenderConfig(samples, CameraBounds(1.8, 6.6), white_bkgd=True, render_num_samples_per_ray=64)

    def model(occupied, colour_of_hat):
        dens = torch.where(occupied, torch.tensor(1.0), torch.tensor(-1.0))[..., None].contiguous()
        feat = torch.zeros(side, side, side, 3)
        feat[..., 0] = 2.0
        feat[hat & ~body] = colour_of_hat
        vg = VoxelGrid(dens, feat.contiguous(), vs, density_preactivation=torch.nn.Identity(),
                       density_postactivation=torch.nn.Softplus(), expected_density_scale=100.0 / 3.0, tunable=True)
        vg.add_attn_params(torch.full_like(dens, -20.0))
        return VolumetricModel(vg, render_sh_voxel_grid, copy.deepcopy(cfg), render_procedure_attn=render_sh_voxel_grid_attn,
                               device=DEV)

    edited = model(body | hat, torch.tensor([-2.0, -2.0, 2.0]))
    reference = model(body, torch.tensor([2.0, 0.0, 0.0]))
    return edited, reference, (hat & ~body), body


class _BlobAttention:
    """stand-in for the UNet cross-attention: token 2 ("hat") lights up where the image is blue, token 1 elsewhere
    on the object; tokens 3.. are flat"""

    def get_num_tokens(self, prompt):
        assert prompt.endswith(" view")
        return 4

    def get_attn_map(self, prompt, pred_rgb, timestamp=0, indices_to_fetch=(7,)):
        assert pred_rgb.dim() == 4 and pred_rgb.shape[1] == 3 and list(indices_to_fetch) == [1, 2, 3, 4]
        rgb = pred_rgb[0]
        blue = (rgb[2] - rgb[0]).clamp(min=0)
        red = (rgb[0] - rgb[2]).clamp(min=0)
        flat = torch.full_like(blue, 0.01)
        return [red, blue, flat, flat], None


def test_refinement_loop_cuts_out_the_hat(tmp_path):
    torch.manual_seed(0)
    np.random.seed(0)
    edited, reference, hat, body = _scene_models()
    vm_edit, vm_obj, vm_out = copy.deepcopy(edited), copy.deepcopy(edited), copy.deepcopy(edited)
    intr = CameraIntrinsics(40, 40, 55.0)
    out = refine_edited_relu_field(
        vm_edit, vm_obj, vm_out, reference, train_dataset=None, hf_auth_token="", output_dir=tmp_path,
        prompt="a ball wearing a hat", edit_idx=[2], timestamp=200, image_dims=None, num_iterations=150,
        learning_rate=0.3, feedback_freq=75, save_freq=75, summary_freq=25, attn_tv_weight=0.001,
        edit_mask_thresh=0.97, num_obj_voxels_thresh=400, min_num_edit_voxels=10, top_k_edit_thresh=30,
        top_k_obj_thresh=30, attn_guidance=_BlobAttention(), camera_intrinsics=intr, camera_bounds=CameraBounds(1.8, 6.6))
    assert out is vm_out
    # both attention grids were optimised (they start at the constant -20)
    for vm in (vm_edit, vm_obj):
        attn = vm.thre3d_repr.attn.detach()
        assert attn.max() > -19.0 and torch.isfinite(attn).all()
    # the learnt edit attention is highest on the hat
    hat_d, body_d = hat.to(DEV), body.to(DEV)
    e_attn = vm_edit.thre3d_repr.attn.detach()[..., 0]
    assert e_attn[hat_d].mean() > e_attn[body_d].mean()
    # the keep grid marks an edit region that lies (mostly) in the hat, and every non-edit voxel was restored
    keep = vm_out.thre3d_repr.attn.detach()[..., 0]
    edit_region = keep == 0
    assert int(edit_region.sum()) > 0
    # (isolated, empty "shell" nodes of the Y-Z dilation also end up on the edit side -- reference behaviour --
    #  so the check looks at occupied voxels)
    occupied = edited.thre3d_repr._densities.detach()[..., 0] > 0
    cut_out = edit_region & occupied
    assert float((cut_out & hat_d).sum()) / float(cut_out.sum()) > 0.8
    assert float((cut_out & hat_d).sum()) / float(hat_d.sum()) > 0.5
    ref_d = reference.thre3d_repr._densities.detach()
    new_d = vm_out.thre3d_repr._densities.detach()
    old_d = edited.thre3d_repr._densities.detach()
    assert torch.equal(new_d[~edit_region], ref_d[~edit_region])
    assert torch.equal(new_d[edit_region], old_d[edit_region])
    assert set(torch.unique(keep).tolist()) <= {0.0, -5.0, -10.0}
    # artefacts
    for name in ("model_final_attn_edit.pth", "model_final_attn_object.pth", "model_final_refined.pth"):
        assert (tmp_path / "saved_models" / name).exists()
    loaded, _ = create_volumetric_model_from_saved_model_attn(tmp_path / "saved_models" / "model_final_refined.pth",
                                                              create_voxel_grid_from_saved_info_dict_attn, device=DEV,
                                                              load_attn=True)
    assert torch.equal(loaded.thre3d_repr._densities.detach(), new_d)
    assert torch.equal(loaded.thre3d_repr.attn.detach()[..., 0], keep)


def test_fused_refinement_step_follows_the_autograd_loop(tmp_path):
    """the loop with one library call per attention grid and iteration (fused_grid_step, voxe_attn_refine_step) against the same
    loop written like the reference (render_rays_attn -> calc_loss_on_attn_grid + TV -> backward -> Adam): same poses, same
    jitter streams, same maps -- the attention grids agree up to float summation order"""
    grids = {}
    for fused in (True, False):
        torch.manual_seed(3)
        np.random.seed(3)
        edited, reference, hat, body = _scene_models(side=32)
        vm_edit, vm_obj, vm_out = copy.deepcopy(edited), copy.deepcopy(edited), copy.deepcopy(edited)
        refine_edited_relu_field(
            vm_edit, vm_obj, vm_out, reference, train_dataset=None, hf_auth_token="", output_dir=tmp_path / str(fused),
            prompt="a ball wearing a hat", edit_idx=[2], timestamp=200, image_dims=None, num_iterations=8,
            learning_rate=0.3, feedback_freq=100, save_freq=100, summary_freq=4, attn_tv_weight=0.01,
            edit_mask_thresh=0.97, num_obj_voxels_thresh=400, min_num_edit_voxels=10, top_k_edit_thresh=30,
            top_k_obj_thresh=30, attn_guidance=_BlobAttention(), camera_intrinsics=CameraIntrinsics(40, 40, 55.0),
            camera_bounds=CameraBounds(1.8, 6.6), fused_grid_step=fused)
        grids[fused] = [vm.thre3d_repr.attn.detach().clone() for vm in (vm_edit, vm_obj)]
    for a, b in zip(grids[True], grids[False]):
        moved = torch.linalg.norm(b + 20.0)
        assert float(moved) > 1.0
        assert float(torch.linalg.norm(a - b) / moved) < 2e-2


def test_get_edit_region_downsampled_and_mismatch_guard():
    torch.manual_seed(1)
    edited, reference, hat, body = _scene_models(side=32)
    vm_edit, vm_obj, vm_out = copy.deepcopy(edited), copy.deepcopy(edited), copy.deepcopy(edited)
    with torch.no_grad():  # hand-made attention grids: edit high in the hat, object high in the body
        vm_edit.thre3d_repr.attn.copy_(torch.where(hat, 3.0, -3.0)[..., None].to(DEV))
        vm_obj.thre3d_repr.attn.copy_(torch.where(body, 3.0, -3.0)[..., None].to(DEV))
    get_edit_region(vm_edit, vm_obj, vm_out, downsample_grid=True, downsample_factor=4, edit_mask_thresh=0.99,
                    num_obj_voxels_thresh=50, min_num_edit_voxels=1, top_k_edit_thresh=5, top_k_obj_thresh=5)
    keep = vm_out.thre3d_repr.attn.detach()[..., 0]
    edit_region = (keep == 0).cpu()
    assert edit_region.any()
    blocks = edit_region.reshape(8, 4, 8, 4, 8, 4).permute(0, 2, 4, 1, 3, 5).reshape(512, 64)
    assert ((blocks.sum(1) == 0) | (blocks.sum(1) == 64)).all()          # whole 4^3 blocks
    assert float((edit_region & (hat | body)).sum()) > 0
    splice_reference_outside_edit_region(vm_out, reference)
    assert torch.equal(vm_out.thre3d_repr._features.detach()[~edit_region.to(DEV)],
                       reference.thre3d_repr._features.detach()[~edit_region.to(DEV)])
    with torch.no_grad():
        vm_obj.thre3d_repr._densities[0, 0, 0, 0] += 1.0
    with pytest.raises(AssertionError, match="Density values"):
        get_edit_region(vm_edit, vm_obj, vm_out)


def test_restore_outside_largest_component():
    side = 24
    dens = torch.full((side, side, side, 1), -1.0)
    dens[2:20, 2:20, 2:20] = 1.0                       # the body (largest)
    rng = np.random.default_rng(3)
    for _ in range(14):                                # 14 isolated floaters of 1..2 voxels
        x, y, z = rng.integers(0, side, 3)
        if not (1 <= x <= 20 and 1 <= y <= 20 and 1 <= z <= 20):
            dens[x, y, z] = 0.5
    dens[21:23, 21:23, 21:23] = 0.7
    dens[0, 0, 23] = dens[23, 0, 0] = dens[0, 23, 0] = dens[23, 23, 0] = dens[0, 23, 23] = 0.4
    dens[23, 0, 23] = dens[23, 23, 23] = dens[12, 0, 0] = dens[0, 12, 0] = dens[0, 0, 12] = 0.4
    feat = torch.zeros(side, side, side, 3)
    vs = VoxelSize(0.1, 0.1, 0.1)
    cfg = SHVoxGridRenderConfig(16, CameraBounds(1.8, 6.6))

    def model(d):
        vg = VoxelGrid(d.clone(), feat.clone(), vs, tunable=True)
        return VolumetricModel(vg, render_sh_voxel_grid, cfg, device=DEV)

    edited, ref = model(dens), model(torch.full_like(dens, -0.25))
    import scipy.ndimage as ndi

    lab, n = ndi.label(dens[..., 0].numpy() > 0, structure=np.ones((3, 3, 3)))
    assert n >= 10
    num = restore_outside_largest_component(edited, ref, k=10)
    assert num == n
    new = edited.thre3d_repr._densities.detach().cpu()[..., 0]
    sizes = np.bincount(lab.ravel())
    largest = lab == (1 + int(np.argmax(sizes[1:])))
    assert torch.equal(new[torch.from_numpy(largest)], dens[..., 0][torch.from_numpy(largest)])
    assert (new[torch.from_numpy(~largest)] == -0.25).all()
    # fewer than k components: no label k exists, everything is restored (reference behaviour)
    few = model(torch.where(torch.from_numpy(largest)[..., None], 1.0, -1.0))
    assert restore_outside_largest_component(few, ref, k=10) == 1
    assert (few.thre3d_repr._densities.detach() == -0.25).all()


def test_segment_entry_point(tmp_path):
    """segment_attn_relu_field.py on checkpoints written to disk: edit / object attention grids with the hat lit up in
    the edit one -> the saved model keeps the edited hat and is the reference field everywhere else"""
    import importlib.util
    import json
    import os

    from click.testing import CliRunner
    from PIL import Image

    from thre3d_atom.data.constants import BOUNDS, EXTRINSIC, FOCAL, HEIGHT, INTRINSIC, ROTATION, TRANSLATION, WIDTH
    from thre3d_atom.modules.volumetric_model import create_volumetric_model_from_saved_model_attn
    from thre3d_atom.thre3d_reprs.voxels import create_voxel_grid_from_saved_info_dict_attn
    from thre3d_atom.utils.constants import CAMERA_BOUNDS, CAMERA_INTRINSICS, HEMISPHERICAL_RADIUS
    from thre3d_atom.utils.imaging_utils import pose_spherical

    torch.manual_seed(0)
    edited, reference, hat, body = _scene_models()
    hat_d, body_d = hat.to(DEV), body.to(DEV)
    intr = CameraIntrinsics(32, 32, 44.0)
    extra = {CAMERA_BOUNDS: CameraBounds(1.8, 6.6), CAMERA_INTRINSICS: intr, HEMISPHERICAL_RADIUS: 4.0311}
    vm_edit, vm_obj = copy.deepcopy(edited), copy.deepcopy(edited)
    # "optimised" attention grids: the edit token lights up the hat, the object token the body
    e = torch.full_like(vm_edit.thre3d_repr.attn.detach(), -4.0)
    e[hat_d] = 3.0
    o = torch.full_like(e, -4.0)
    o[body_d] = 3.0
    vm_edit.thre3d_repr.add_attn_params(e)
    vm_obj.thre3d_repr.add_attn_params(o)
    paths = {}
    plain_ref, plain_sds = copy.deepcopy(reference), copy.deepcopy(edited)
    for vm in (plain_ref, plain_sds):          # the reconstruction / SDS stages save fields without an attention grid
        del vm.thre3d_repr.attn
        vm.thre3d_repr.attn = None
    for name, vm in (("ref", plain_ref), ("sds", plain_sds), ("edit", vm_edit), ("obj", vm_obj)):
        paths[name] = tmp_path / f"{name}.pth"
        torch.save(vm.get_save_info(extra), paths[name])
    data = tmp_path / "data"
    (data / "train").mkdir(parents=True)
    params = {}
    for i in range(2):
        pose = pose_spherical(120.0 * i, 30.0, 4.0311)
        Image.fromarray(np.zeros((32, 32, 3), np.uint8)).save(data / "train" / f"r_{i}.png")
        params[f"r_{i}.png"] = {EXTRINSIC: {ROTATION: pose.rotation.numpy().tolist(), TRANSLATION: pose.translation.numpy().tolist()},
                                INTRINSIC: {HEIGHT: 32, WIDTH: 32, FOCAL: 44.0, BOUNDS: [2.0, 6.0]}}
    (data / "train_camera_params.json").write_text(json.dumps(params))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("segment_cli", os.path.join(root, "segment_attn_relu_field.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    res = CliRunner().invoke(mod.main, ["-d", str(data), "-ie", str(paths["edit"]), "-io", str(paths["obj"]), "-o", str(out),
                                        "-r", str(paths["ref"]), "-i", str(paths["sds"]), "--edit_mask_thresh", "0.97",
                                        "--num_obj_voxels_thresh", "400", "--min_num_edit_voxels", "10",
                                        "--top_k_edit_thresh", "30", "--top_k_obj_thresh", "30", "--log_wandb", "False"])
    assert res.exit_code == 0, (res.output, res.exception)
    assert (out / "training_logs" / "rendered_output" / "sds_refined_0.png").exists()
    vm, saved_extra = create_volumetric_model_from_saved_model_attn(
        out / "saved_models" / "model_final_refined.pth", create_voxel_grid_from_saved_info_dict_attn, device=DEV, load_attn=True)
    assert abs(saved_extra[HEMISPHERICAL_RADIUS] - 4.0311) < 1e-3
    keep = vm.thre3d_repr.attn.detach()[..., 0]
    region = keep == 0
    occupied = edited.thre3d_repr._densities.detach()[..., 0] > 0
    cut = region & occupied
    assert int(cut.sum()) > 0 and float((cut & hat_d).sum()) / float(cut.sum()) > 0.8
    new_d, new_f = vm.thre3d_repr._densities.detach(), vm.thre3d_repr._features.detach()
    ref_d, ref_f = reference.thre3d_repr._densities.detach(), reference.thre3d_repr._features.detach()
    sds_d = edited.thre3d_repr._densities.detach()
    assert torch.equal(new_d[~region], ref_d[~region]) and torch.equal(new_f[~region], ref_f[~region])
    assert torch.equal(new_d[region], sds_d[region])
