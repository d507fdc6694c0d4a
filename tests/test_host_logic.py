"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/voxe.h
declares, descriptor construction, API containers, config override rules, checkpoint loading.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from voxe_hip import abi
from voxe_hip.desc import make_grid_desc, make_render_cfg, norm_constants


def _declared(prefix):
    text = open(os.path.join(ROOT, "include", "voxe.h")).read()
    names = set(re.findall(r"\b(voxe_[a-z0-9_]+)\s*\(", text))
    if prefix == "voxe_cpu_":
        return sorted(n for n in names if n.startswith("voxe_cpu_"))
    return sorted(n for n in names if not n.startswith("voxe_cpu_"))


def test_hip_library_exports_every_declared_symbol():
    from voxe_hip import build

    path = build.build()
    handle = ctypes.CDLL(path)
    declared = _declared("voxe_")
    assert set(declared) == set(abi.hip_symbols()), "abi.py and voxe.h disagree"
    for name in declared:
        assert hasattr(handle, name), f"{name} missing from libvoxe_hip.so"
    abi.declare(handle, "voxe_")
    assert handle.voxe_abi_version() == abi.ABI_VERSION
    assert handle.voxe_strerror(-4).decode().startswith("workspace")
    # argument validation happens before any device work
    g = make_grid_desc(0, 0, (4, 4, 4), 3, [(-1, 1)] * 3, 1.0, abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    c = make_render_cfg(8, 1.0, 2.0)
    assert handle.voxe_render_fwd(ctypes.byref(g), ctypes.byref(c), None, None, 4, None, None, None, None, None, None, 0, None) == abi.ERR_NULL_POINTER
    g.densities, g.features = 8, 8
    g.F = 5
    assert handle.voxe_render_fwd(ctypes.byref(g), ctypes.byref(c), 8, 8, 4, None, 8, None, None, None, 8, 1 << 20, None) == abi.ERR_BAD_SHAPE
    g.F = 3
    g.density_post_act = 9
    assert handle.voxe_render_fwd(ctypes.byref(g), ctypes.byref(c), 8, 8, 4, None, 8, None, None, None, 8, 1 << 20, None) == abi.ERR_UNSUPPORTED
    g.density_post_act = abi.ACT_RELU
    assert handle.voxe_render_fwd(ctypes.byref(g), ctypes.byref(c), 8, 8, 4, None, 8, None, None, None, None, 0, None) == abi.ERR_WORKSPACE
    assert handle.voxe_workspace_bytes(ctypes.byref(g), ctypes.byref(c), 4) >= 2 * 4 * 4 * 4 * 4 * 4


def test_oracle_library_exports_every_declared_symbol():
    from oracle import voxe_oracle as vo

    handle = ctypes.CDLL(vo.build())
    declared = _declared("voxe_cpu_")
    assert set(declared) == set(abi.cpu_symbols())
    for name in declared:
        assert hasattr(handle, name)


def test_struct_layout_matches_header():
    """ctypes mirror vs the C compiler's view of include/voxe.h (sizes and a few offsets)."""
    import subprocess
    import tempfile

    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "voxe.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(VoxeAttnRefineStep), offsetof(VoxeAttnRefineStep, tv_loss_always),
             offsetof(VoxeAttnRefineStep, step), offsetof(VoxeAttnRefineStep, exp_avg_sq), offsetof(VoxeAttnRefineStep, zero_gradient_first),
             sizeof(VoxeReconStep), offsetof(VoxeReconStep, losses));
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(VoxeGridDesc), offsetof(VoxeGridDesc, aabb_lo),
             offsetof(VoxeGridDesc, density_scale), sizeof(VoxeRenderCfg), offsetof(VoxeRenderCfg, seed),
             offsetof(VoxeRenderCfg, reuse_packed_grid), offsetof(VoxeRenderCfg, image_width),
             offsetof(VoxeRenderCfg, ray_state_valid), offsetof(VoxeRenderCfg, dispatch), sizeof(VoxeDispatch),
             offsetof(VoxeDispatch, tile_min_rays), offsetof(VoxeDispatch, fwd_window), offsetof(VoxeDispatch, region_image_ratio));
      return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "t.c")
        open(p, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    G, R, D = abi.VoxeGridDesc, abi.VoxeRenderCfg, abi.VoxeDispatch
    A, RS = abi.VoxeAttnRefineStep, abi.VoxeReconStep          # (ABI v10 / v7: the per-iteration library calls)
    assert vals[:7] == [ctypes.sizeof(A), A.tv_loss_always.offset, A.step.offset, A.exp_avg_sq.offset, A.zero_gradient_first.offset,
                        ctypes.sizeof(RS), RS.losses.offset]
    vals = vals[7:]
    assert vals == [ctypes.sizeof(G), G.aabb_lo.offset, G.density_scale.offset, ctypes.sizeof(R), R.seed.offset,
                    R.reuse_packed_grid.offset, R.image_width.offset, R.ray_state_valid.offset, R.dispatch.offset,
                    ctypes.sizeof(D), D.tile_min_rays.offset, D.fwd_window.offset, D.region_image_ratio.offset]


def test_dispatch_is_resolved_from_the_environment_once_and_travels_per_call(monkeypatch):
    """VoxeDispatch (ABI v7): the library reads no environment on the render path; the binding turns the VOXE_* switches into
    ONE struct on first use, later changes of the environment are not seen, and a call can carry its own dispatch"""
    import dataclasses

    from voxe_hip import dispatch as dp

    dp.from_env.cache_clear()
    for k, v in {"VOXE_TILE_MIN_RAYS": "0", "VOXE_REGION_MIN_RAYS": "-1", "VOXE_FWD_TILE": "0", "VOXE_TILE_KL": "10",
                 "VOXE_BWD_MODE": "packed", "VOXE_REGION_IMAGE_RATIO": "0", "VOXE_TILE_FIT_M": "4.5", "VOXE_TILE_MAP": "rows"}.items():
        monkeypatch.setenv(k, v)
    try:
        d = dp.from_env()
        assert (d.tile_min_rays, d.region_min_rays, d.fwd_window, d.tile_kl, d.bwd_mode, d.region_image_ratio, d.tile_fit_m,
                d.tile_map) == (-1, -1, -1, 10, 2, -1.0, 4.5, 3)
        monkeypatch.setenv("VOXE_TILE_KL", "8")
        assert dp.from_env().tile_kl == 10 and dp.current() is d          # resolved once
        with dp.override(tile_kl=8) as o:
            assert dp.current() is o and o.tile_kl == 8 and o.bwd_mode == 2
        assert dp.current() is d
        st = d.struct()
        assert st is d.struct() and (st.tile_min_rays, st.region_min_rays, st.tile_kl) == (-1, -1, 10)
        c = make_render_cfg(8, 1.0, 2.0, dispatch=st)
        assert c.dispatch.contents.tile_kl == 10 and not make_render_cfg(8, 1.0, 2.0).dispatch   # NULL = shipped
        assert dataclasses.asdict(dp.SHIPPED) == {f.name: 0 for f in dataclasses.fields(dp.Dispatch)}
    finally:
        for k in list(os.environ):
            if k.startswith("VOXE_"):
                monkeypatch.delenv(k, raising=False)
        dp.from_env.cache_clear()
    # no source file of the library reads the environment on the render path (two debug hooks of the graph cut remain)
    csrc = os.path.join(ROOT, "vox-e_amd", "csrc")
    hits = [(f, l.strip()) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".hpp"))
            for l in open(os.path.join(csrc, f)) if "getenv" in l]
    assert len(hits) <= 2 and all(f == "voxe_refine.hip" for f, _ in hits), hits


def test_norm_constants_are_float32_like_reference():
    scale, bias = norm_constants([(-1.5, 1.5), (-0.775, 0.775), (0.1, 0.9)])
    for (lo, hi), s, b in zip([(-1.5, 1.5), (-0.775, 0.775), (0.1, 0.9)], scale, bias):
        es = (np.float32(1) - np.float32(-1)) / (np.float32(hi) - np.float32(lo))
        assert s == es and b == np.float32(-1) - np.float32(lo) * es and s.dtype == np.float32


def test_api_containers_and_config_rules():
    from thre3d_atom.modules.volumetric_model import VolumetricModel
    from thre3d_atom.rendering.volumetric.render_interface import Rays, RenderOut, RenderOutAttn
    from thre3d_atom.thre3d_reprs.renderers import SHVoxGridRenderConfig
    from thre3d_atom.utils.imaging_utils import CameraBounds

    r = Rays(torch.zeros(10, 3), torch.ones(10, 3), image_shape=(2, 5))
    assert len(r) == 10 and len(r[2:5]) == 3 and r[2:5].image_shape is None
    with pytest.raises(AssertionError):
        Rays(torch.zeros(10, 3), torch.zeros(9, 3))
    with pytest.raises(AssertionError):
        Rays(torch.zeros(10, 2), torch.zeros(10, 2))
    out = RenderOut(torch.zeros(4, 3, requires_grad=True), torch.zeros(4, 1))
    assert out.extra == {} and not out.detach().colour.requires_grad
    with pytest.raises(AssertionError):
        RenderOut(torch.zeros(4, 2), torch.zeros(4, 1))
    with pytest.raises(AssertionError):
        RenderOutAttn(torch.zeros(4, 3), torch.zeros(4, 1))
    cfg = SHVoxGridRenderConfig(64, CameraBounds(1.0, 2.0))
    assert cfg.perturb_sampled_points and cfg.parallel_rays_chunk_size == 32768 and cfg.render_num_samples_per_ray == 1024
    upd = VolumetricModel._update_render_config(cfg, {"white_bkgd": True, "num_samples_per_ray": 8})
    assert upd.white_bkgd and upd.num_samples_per_ray == 8 and not cfg.white_bkgd
    with pytest.raises(ValueError):
        VolumetricModel._update_render_config(cfg, {"no_such_field": 1})


def test_voxel_grid_geometry_and_activation_mapping():
    from thre3d_atom.thre3d_reprs.voxels import VoxelGrid, VoxelGridLocation, VoxelSize, density_activation_codes
    from voxe_hip.runtime import VoxeError

    vg = VoxelGrid(torch.zeros(5, 6, 7, 1), torch.zeros(5, 6, 7, 3), VoxelSize(0.31, 0.27, 0.23),
                   VoxelGridLocation(0.5, 0.0, -0.25), tunable=True)
    assert vg.grid_dims == (5, 6, 7)
    assert vg.aabb.x_range == (0.5 - 5 * 0.31 / 2, 0.5 + 5 * 0.31 / 2)
    assert set(vg.state_dict().keys()) == {"_densities", "_features"}
    pts = torch.tensor([[0.5, 0.0, -0.25], [10.0, 0.0, 0.0], [vg.aabb.x_range[0], 0.0, -0.25]])
    assert vg.test_inside_volume(pts)[:, 0].tolist() == [True, False, False]
    assert density_activation_codes(torch.abs, torch.nn.Identity()) == (abi.ACT_ABS, abi.ACT_IDENTITY)
    assert density_activation_codes(torch.nn.Identity(), torch.nn.Softplus()) == (abi.ACT_IDENTITY, abi.ACT_SOFTPLUS)
    assert density_activation_codes(torch.nn.Identity(), torch.nn.ReLU()) == (abi.ACT_IDENTITY, abi.ACT_RELU)
    with pytest.raises(VoxeError):
        density_activation_codes(torch.nn.Identity(), torch.nn.Softplus(beta=2))
    with pytest.raises(VoxeError):
        density_activation_codes(torch.tanh, torch.nn.Identity())
    spec = vg.voxe_grid_spec()
    assert spec.density_pre_act == abi.ACT_ABS and spec.feature_kind == abi.FEAT_SH


def test_reference_checkpoint_loads_and_product_refuses_cpu():
    from thre3d_atom.modules.volumetric_model import create_volumetric_model_from_saved_model
    from thre3d_atom.rendering.volumetric.render_interface import Rays
    from thre3d_atom.thre3d_reprs.renderers import render_sh_voxel_grid
    from thre3d_atom.thre3d_reprs.voxels import (
        create_voxel_grid_from_saved_info_dict,
        create_voxel_grid_from_saved_info_dict_attn,
    )
    from voxe_hip.runtime import VoxeError

    path = os.path.join(GOLDEN, "ref_checkpoint.pth")
    vm, extra = create_volumetric_model_from_saved_model(path, create_voxel_grid_from_saved_info_dict)
    assert vm.render_procedure is render_sh_voxel_grid  # unpickled by qualified name onto this package
    assert vm.thre3d_repr.grid_dims == (6, 6, 6) and extra["hemispherical_radius"] == 4.0311
    data = torch.load(path, weights_only=False)
    vg = create_voxel_grid_from_saved_info_dict_attn(data)
    assert vg.attn is not None and float(vg.attn.mean()) == -20.0
    # save -> load round trip of our own checkpoint
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        p2 = os.path.join(td, "m.pth")
        torch.save(vm.get_save_info(extra), p2)
        vm2, _ = create_volumetric_model_from_saved_model(p2, create_voxel_grid_from_saved_info_dict)
        assert torch.equal(vm2.thre3d_repr.densities, vm.thre3d_repr.densities)
    if not torch.cuda.is_available():
        # no silent CPU fallback: the product path fails loudly without a GPU
        rays = Rays(torch.zeros(4, 3), torch.ones(4, 3))
        with pytest.raises(VoxeError):
            vm.render_rays(rays)


def test_per_iteration_entry_points_refuse_cpu_tensors():
    """the r05 entry points (voxe_attn_refine_step, voxe_attn_masked_l1) have no CPU path either: CPU tensors raise VoxeError
    before anything is launched -- on a box without a GPU as well as on one with"""
    from voxe_hip import ops
    from voxe_hip.runtime import VoxeError

    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=3.0, feature_kind=abi.FEAT_ATTN)
    params = ops.RenderParams(num_samples=8, near=1.8, far=6.6, image_width=4)
    dens, attn = torch.zeros(4, 4, 4, 1), torch.zeros(4, 4, 4, 1)
    rays = torch.zeros(16, 3)
    state = (torch.zeros_like(attn), torch.zeros_like(attn))
    with pytest.raises(VoxeError):
        ops.attn_refine_step_(spec, params, dens, attn, rays, rays + 1.0, torch.zeros(16), ops.Workspace(), 1, 0.01, state, 0.0)
    with pytest.raises(VoxeError):
        ops.attn_masked_l1(torch.zeros(16, 1), torch.zeros(4, 4))
    assert attn.abs().sum() == 0 and state[0].abs().sum() == 0
    # ... and every entry point that asks for the device first says the same (not a bare RuntimeError from torch.cuda)
    with pytest.raises(VoxeError):
        ops.grid_adam_step_(spec, dens, attn, abi.GRAD_LINEAR, ops.Workspace(), 1, 0.01, state_features=state)
    with pytest.raises(VoxeError):
        ops.render_fwd_into(spec, params, dens, attn, rays, rays + 1.0, None, *[torch.zeros(16, 1) for _ in range(4)], ops.Workspace())


def test_datasets_on_disk_format_and_downsampling(tmp_path):
    """`*_camera_params.json` + image folder (reference data/datasets.py, data/constants.py) round trip."""
    import json

    from PIL import Image

    from thre3d_atom.data.datasets import InMemoryPosedImages, PosedImagesDataset
    from thre3d_atom.utils.imaging_utils import CameraBounds, CameraIntrinsics, pose_spherical

    img_dir = tmp_path / "train"
    img_dir.mkdir()
    params = {}
    rng = np.random.default_rng(0)
    for i in range(3):
        pose = pose_spherical(40.0 * i, 30.0, 4.0311)
        rgba = (rng.random((12, 16, 4)) * 255).astype(np.uint8)
        Image.fromarray(rgba, "RGBA").save(img_dir / f"r_{i}.png")
        params[f"r_{i}.png"] = {
            "extrinsic": {"rotation": pose.rotation.numpy().tolist(), "translation": pose.translation.numpy().tolist()},
            "intrinsic": {"height": 12, "width": 16, "focal": 20.0, "bounds": [2.0, 6.0]},
        }
    (tmp_path / "train_camera_params.json").write_text(json.dumps(params))
    ds = PosedImagesDataset(img_dir, tmp_path / "train_camera_params.json", rgba_white_bkgd=True)
    assert len(ds) == 3 and ds.images.shape == (3, 3, 12, 16) and ds.poses.shape == (3, 3, 4)
    assert ds.camera_intrinsics == CameraIntrinsics(12, 16, 20.0)
    # datasets.py:275-276, evaluated in float32 like the reference (np.float32 bounds array * python float)
    assert ds.camera_bounds == CameraBounds(float(np.float32(2.0) * 0.9), float(np.float32(6.0) * 1.1))
    assert abs(ds.get_hemispherical_radius_estimate() - 4.0311) < 1e-3       # test_datasets.py:48-52
    assert 0.0 <= float(ds.images.min()) and float(ds.images.max()) <= 1.0
    img, pose, idx = ds[1]
    assert img.shape == (3, 12, 16) and idx == 1 and abs(float(torch.det(pose[:, :3])) - 1.0) < 1e-4
    half = ds.downsampled(2.0)
    assert half.images.shape == (3, 3, 6, 8) and half.camera_intrinsics == CameraIntrinsics(6, 8, 10.0)
    assert isinstance(half, InMemoryPosedImages)
    # directional views (a direction prompt word per camera, reference datasets.py:41,85-88,331-335,387-390)
    for i, word in enumerate(("front", "side", "back")):
        params[f"r_{i}.png"]["dir"] = word
    (tmp_path / "train_camera_params.json").write_text(json.dumps(params))
    dd = PosedImagesDataset(img_dir, tmp_path / "train_camera_params.json", rgba_white_bkgd=True, directional=True)
    img, pose, direction, idx = dd[2]
    assert direction == "back" and idx == 2 and img.shape == (3, 12, 16)
    assert PosedImagesDataset.extract_dir(dd.camera_parameters["r_1.png"]) == "side"
    cam = PosedImagesDataset.extract_pose(dd.camera_parameters["r_1.png"])
    assert cam.rotation.shape == (3, 3) and cam.translation.shape == (3, 1)
    assert dd.get_config_dict()["rgba_white_bkgd"] is True and dd.get_config_dict()["downsample_factor"] == 1.0


def test_dataset_bounds_scale_and_truncated_intrinsics_follow_the_reference(tmp_path):
    """reference data/datasets.py: bounds = (min over ALL cameras' near) * 0.9, (max far) * 1.1 (:267-277); with
    normalize_scene_scale every location and both bounds are divided by the distance of the FARTHEST camera (:218-249);
    a downsample factor divides (height, width, focal) with height / width truncated (:288-306): 800 / 3 -> 266, f / 3"""
    import json

    from PIL import Image

    from thre3d_atom.data.datasets import PosedImagesDataset
    from thre3d_atom.utils.imaging_utils import CameraIntrinsics, pose_spherical

    img_dir = tmp_path / "train"
    img_dir.mkdir()
    params = {}
    radii, bounds = (3.0, 5.0, 4.0), ([2.0, 6.0], [1.5, 5.0], [2.5, 7.0])
    for i in range(3):
        pose = pose_spherical(50.0 * i, 25.0, radii[i])
        Image.fromarray(np.full((80, 80, 3), 40 * i, np.uint8), "RGB").save(img_dir / f"r_{i}.png")
        params[f"r_{i}.png"] = {
            "extrinsic": {"rotation": pose.rotation.numpy().tolist(), "translation": pose.translation.numpy().tolist()},
            "intrinsic": {"height": 80, "width": 80, "focal": 100.0, "bounds": bounds[i]},
        }
    (tmp_path / "p.json").write_text(json.dumps(params))
    ds = PosedImagesDataset(img_dir, tmp_path / "p.json")
    assert abs(ds.camera_bounds.near - 1.5 * 0.9) < 1e-6 and abs(ds.camera_bounds.far - 7.0 * 1.1) < 1e-6
    dn = PosedImagesDataset(img_dir, tmp_path / "p.json", normalize_scene_scale=True)
    norms = dn.poses[:, :, 3].norm(dim=-1)
    assert abs(float(norms.max()) - 1.0) < 1e-6 and abs(float(norms.min()) - 3.0 / 5.0) < 1e-6
    assert abs(dn.camera_bounds.near - 1.5 * 0.9 / 5.0) < 1e-6 and abs(dn.camera_bounds.far - 7.0 * 1.1 / 5.0) < 1e-6
    d3 = PosedImagesDataset(img_dir, tmp_path / "p.json", downsample_factor=3.0)
    assert d3.camera_intrinsics == CameraIntrinsics(26, 26, 100.0 / 3.0) and d3.images.shape[-2:] == (26, 26)


def test_edit_stage_intrinsics_follow_the_dataset_or_the_downsample_factor():
    """the SDS edit renders at the dataset's intrinsics (built at --data_downsample_factor, default 3.0), not
    at the checkpoint's training resolution; in uncoupled_mode rendered and target pixel counts must agree"""
    import importlib.util
    import os
    import types

    from thre3d_atom.utils.imaging_utils import CameraIntrinsics

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("edit_cli", os.path.join(root, "edit_pretrained_relu_field.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ckpt = CameraIntrinsics(800, 800, 1111.111)
    got = mod.edit_stage_intrinsics(ckpt, None, 3.0)
    assert (got.height, got.width) == (266, 266) and abs(got.focal - 1111.111 / 3.0) < 1e-9
    assert mod.edit_stage_intrinsics(ckpt, None, 1.0) == ckpt
    ds = types.SimpleNamespace(camera_intrinsics=CameraIntrinsics(266, 266, 370.37))
    assert mod.edit_stage_intrinsics(ckpt, ds, 3.0) is ds.camera_intrinsics   # rays and target pixels share these


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under vox-e_amd/ nor the entry scripts may import / load it, and
    importing the whole product package must not pull it in"""
    import ast
    import glob
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "vox-e_amd", "**", "*.py"), recursive=True)
    files += [os.path.join(root, n) for n in ("render_sh_based_voxel_grid.py", "render_sh_based_voxel_grid_attn.py", "segment_attn_relu_field.py", "edit_pretrained_relu_field.py",
                                              "refine_edited_relu_field.py",
                                              "train_sh_based_voxel_grid_with_posed_images.py")]
    for path in files:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] == "oracle" or "voxe_oracle" in n for n in names), path
        if not path.endswith(os.path.join("voxe_hip", "abi.py")):  # (its docstring names both libraries the ABI describes)
            assert "libvoxe_oracle" not in open(path).read(), path
    code = ("import sys; sys.path.insert(0, %r); import voxe_hip.ops, thre3d_atom.modules.sds_trainer, "
            "thre3d_atom.modules.trainers, thre3d_atom.modules.attn_grid_trainer, thre3d_atom.modules.refinement_functions; "
            "bad = [m for m in sys.modules if m.split('.')[0] == 'oracle' or 'voxe_oracle' in m]; "
            "assert not bad, bad" % os.path.join(root, "vox-e_amd"))
    subprocess.check_call([sys.executable, "-c", code])


def test_entry_points_accept_every_reference_option():
    """the reference's shell scripts must run unchanged: every option its four entry points declare
    (tests/golden/cli_options.json, tools/gen_cli_options.py) is accepted by this build's entry point of the same name"""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "cli_options.json")))
    for script, options in ref.items():
        spec = importlib.util.spec_from_file_location("entry_" + script[:-3], os.path.join(root, script))
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        params = {o: p for p in module.main.params for o in p.opts}
        missing = sorted(set(options) - set(params))
        assert not missing, (script, missing)
        for name, info in options.items():     # short aliases too (-i, -o, -p ...)
            for alias in info["names"]:
                assert alias in params or alias in {o for p in module.main.params for o in p.secondary_opts}, (script, alias)
            if info["nargs"]:
                assert params[name].nargs == info["nargs"], (script, name)


def test_entry_point_defaults_equal_the_reference():
    """options that exist in the reference keep the reference's default values"""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "cli_options.json")))
    for script, options in ref.items():
        spec = importlib.util.spec_from_file_location("entry2_" + script[:-3], os.path.join(root, script))
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        params = {o: p for p in module.main.params for o in p.opts}
        for name, info in options.items():
            if info["default"] is None:
                continue
            try:
                want = eval(info["default"], {}, {})      # literals only ("5000 * 100", "(3.0, 3.0, 3.0)", '"Vox-E"')
            except Exception:
                continue
            got = params[name].default
            got = tuple(got) if isinstance(want, tuple) else got
            assert got == want, (script, name, got, want)


def test_nerf_blender_converter_round_trip(tmp_path):
    """tools/convert_from_nerf_blender_dataset.py (reference tools/convert_from_nerf_blender_dataset.py:33-90): a tiny
    NeRF-synthetic scene -> <split>_camera_params.json -> PosedImagesDataset gives back the poses, the focal length from
    camera_angle_x and the widened [2, 6] bounds"""
    import importlib.util
    import json

    from click.testing import CliRunner
    from PIL import Image

    from thre3d_atom.data.datasets import PosedImagesDataset
    from thre3d_atom.utils.imaging_utils import pose_spherical

    src = tmp_path / "lego"
    frames = []
    (src / "train").mkdir(parents=True)
    poses = []
    for i in range(3):
        pose = pose_spherical(50.0 * i, 25.0, 4.0311)
        m = np.eye(4, dtype=np.float64)
        m[:3, :3], m[:3, 3:] = pose.rotation.numpy(), pose.translation.numpy()
        poses.append(m)
        Image.fromarray((np.random.default_rng(i).random((10, 14, 3)) * 255).astype(np.uint8)).save(src / "train" / f"r_{i}.png")
        frames.append({"file_path": f"./train/r_{i}", "rotation": 0.01, "transform_matrix": m.tolist()})
    (src / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911112, "frames": frames}))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("convert_cli", os.path.join(root, "tools", "convert_from_nerf_blender_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "converted"
    res = CliRunner().invoke(mod.main, ["-d", str(src), "-o", str(out), "--link_images", "True"])
    assert res.exit_code == 0, (res.output, res.exception)
    assert not (out / "val_camera_params.json").exists()            # absent splits are skipped
    ds = PosedImagesDataset(out / "train", out / "train_camera_params.json")
    assert len(ds) == 3 and ds.images.shape == (3, 3, 10, 14)
    assert abs(ds.camera_intrinsics.focal - 0.5 * 14 / np.tan(0.5 * 0.6911112)) < 1e-4
    assert abs(ds.camera_bounds.near - 1.8) < 1e-6 and abs(ds.camera_bounds.far - 6.6) < 1e-6
    for i in range(3):
        np.testing.assert_allclose(ds.poses[i].numpy(), poses[i][:3, :4].astype(np.float32), atol=1e-6)


def test_newest_pmc_summary_belongs_to_these_kernel_sources():
    """bench.py combines counter values only with timings of the kernels they were collected on: the newest committed
    profiles/*_pmc_summary.json must carry the hash of THIS tree's vox-e_amd/csrc/* + include/voxe.h (tools/pmc_to_json.py
    stamps it; a kernel edit without a fresh PMC run makes the bench line say "stale" -- and fails here)"""
    import json
    import re

    from voxe_hip.build import source_hash

    h = source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == source_hash()
    prof = os.path.join(ROOT, "profiles")
    newest = sorted(f for f in os.listdir(prof) if f.endswith("_pmc_summary.json"))[-1]
    summary = json.load(open(os.path.join(prof, newest)))
    assert re.fullmatch(r"[0-9a-f]{16}", summary.get("source_hash", "")), "PMC summaries must be stamped with the hash of their kernel sources"
    if summary["source_hash"] != h:
        # a legitimate state between a kernel edit and the next PMC run: bench.py then reports {"stale": true} instead of
        # counter-derived fractions (checked on the GPU by tests/test_bench_two_ranks_gpu.py); say so loudly, do not hide it
        import warnings

        warnings.warn(f"{newest} was collected on other kernel sources ({summary['source_hash']} != {h}): "
                      f"re-run tools/gpu_pmc.sh + tools/pmc_to_json.py before quoting roofline.physical")
    # the headline backward: the lean kernel since r05, the general LDS-window kernel in older summaries
    key = [k for k in summary["kernels"] if k.startswith(("voxe::render_bwd_tile4_kernel<8, false, 0>",
                                                            "voxe::render_bwd_tile_kernel<3, 1, 1, true, true, 0, 8"))]
    assert key and {"FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE"} <= set(summary["kernels"][key[0]])


def test_isa_issue_model_prices_instructions_like_the_microbenchmark():
    """tools/isa_issue_model.py (r06): the class a VALU instruction is priced in follows profiles/r06_valu_rate.txt -- double rate only
    for the plain f32 / integer forms WITHOUT an SGPR source, DPP or SDWA; v_cndmask_b32 behind a scalar write of vcc is the 22.8-clk
    case; the committed profile names the two headline kernels with the sample loops found in their assembly"""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("isa_issue_model", os.path.join(ROOT, "tools", "isa_issue_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.classify("v_fma_f32", "v1, v2, v3, v4", "V") == "fast"
    assert m.classify("v_fma_f32", "v1, v2, s20, v4", "V") == "slow"            # one SGPR source: 4.27 clk measured
    assert m.classify("v_mul_f32_e32", "v1, 0x3f800000, v2", "V") == "fast"     # literals / inline constants stay double rate
    assert m.classify("v_add_f32_dpp", "v1, v2, v3 row_shr:1 row_mask:0xf bank_mask:0xf", "V") == "slow"
    assert m.classify("v_lshlrev_b32_e32", "v1, 4, v2", "V") == "slow" and m.classify("v_lshrrev_b32_e32", "v1, 4, v2", "V") == "fast"
    assert m.classify("v_exp_f32_e32", "v1, v2", "V") == "trans" and m.classify("v_mul_f64", "v[0:1], v[2:3], v[4:5]", "V") == "slow"
    assert m.classify("v_cndmask_b32_e32", "v1, v2, v3, vcc", "S") == "cnd_salu_vcc"
    assert m.classify("v_cndmask_b32_e32", "v1, v2, v3, vcc", "V") == "slow"
    assert m.classify("v_cndmask_b32_e64", "v1, v2, v3, s[4:5]", "S") == "slow"
    assert m.pmc_class("v_mul_f64") == "MUL_F64" and m.pmc_class("v_cvt_f64_f32_e32") == "CVT" and m.pmc_class("v_exp_f32_e32") == "TRANS_F32"
    assert m.pmc_class("v_fmac_f32_e32") == "FMA_F32" and m.pmc_class("v_and_b32_e32") == "OTHER"
    blocks = m.price(m.blocks_of("k:\n.LBB0_1:\n\ts_and_b64 vcc, s[0:1], s[2:3]\n\tv_cndmask_b32_e32 v1, v2, v3, vcc\n\tv_cmp_lt_f32_e32 vcc, v1, v2\n"
                                 "\tv_cndmask_b32_e32 v1, v2, v3, vcc\n\tv_fma_f32 v1, v1, v2, v3\n\ts_cbranch_scc1 .LBB0_1\n"))
    assert dict(blocks[1]["cls"]) == {"cnd_salu_vcc": 1, "slow": 2, "fast": 1} and m.loops_of(blocks) == [(1, 1)]
    prof = json.load(open(os.path.join(ROOT, "profiles", "r06_issue_model.json")))
    assert prof["cost_table_clk"] == m.COST
    for key in ("voxe::render_bwd_tile4_kernel<8, false, 0>", "voxe::render_fwd_tile4w_kernel<false>"):
        k = prof["kernels"][key]
        assert 2.15 < k["clk_per_valu"] < 4.3 and k["hot_loops"] and k["occupancy_waves_per_simd"] >= 3


def test_pinned_staging_draws_the_generators_stream_and_reuses_its_buffers():
    """trainers._PinnedStaging (r06): the per-iteration camera picks go through a small ring of staging buffers (pinned on a GPU box, so
    that the copy does not wait for the stream); the values are exactly the generator's draws, in order, whatever the ring's depth"""
    import torch

    from thre3d_atom.modules.trainers import _PinnedStaging

    gen, ref = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    st = _PinnedStaging(6, "cpu", depth=2)
    outs = [st.to_device(lambda out: torch.randint(0, 100, (out.numel(),), generator=gen, out=out)) for _ in range(5)]
    for o in outs:
        assert torch.equal(o, torch.randint(0, 100, (6,), generator=ref))
    assert outs[0].data_ptr() != outs[2].data_ptr()       # (what the caller gets is its own tensor, not the ring's buffer)
