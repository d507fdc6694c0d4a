"""The drop-in boundary from the other side: tests/c_abi/voxe_c_consumer.c -- plain C, compiled with gcc against include/voxe.h,
libvoxe_hip.so and the HIP runtime's C API, no Python / torch in the process -- renders forward + backward; its outputs equal the
Python binding's for the same inputs (the same kernels: forward bit for bit, gradients up to the order of the float atomics)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from synth import RADIUS, focal_for, synth_pose_angles
from voxe_hip import abi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_matches_the_python_binding(tmp_path):
    from thre3d_atom.utils.imaging_utils import pose_spherical
    from voxe_hip import build, ops

    lib_path = build.build()
    exe = str(tmp_path / "voxe_c_consumer")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "voxe_c_consumer.c"), "-o", exe, lib_path,
                           "-L", os.path.join(rocm, "lib"), "-lamdhip64", f"-Wl,-rpath,{os.path.dirname(lib_path)}",
                           f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"])
    dims, hw, S = (24, 20, 28), 96, 64
    rng = np.random.default_rng(0)
    dens = rng.uniform(-1, 1, (*dims, 1)).astype(np.float32)
    feat = rng.uniform(-1, 1, (*dims, 3)).astype(np.float32)
    gc = rng.standard_normal((hw * hw, 3)).astype(np.float32)
    pose = pose_spherical(*synth_pose_angles(5, 100), RADIUS)
    focal = np.float32(focal_for(hw))
    rot, trans = pose.rotation.numpy().astype(np.float32), pose.translation.numpy().astype(np.float32).reshape(3)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([*dims, hw, hw, S], np.int32).tofile(f)
        np.concatenate([[focal], rot.reshape(-1), trans]).astype(np.float32).tofile(f)
        dens.tofile(f); feat.tofile(f); gc.tofile(f)
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "voxe C consumer ok" in res.stdout
    raw = np.fromfile(tmp_path / "out.bin", np.float32)
    R, nvox = hw * hw, int(np.prod(dims))
    parts = np.split(raw, np.cumsum([3 * R, R, R, nvox]))
    c_col, c_dep, c_acc, c_dd, c_df = parts
    # the Python binding on the same inputs
    dev = torch.device("cuda:0")
    spec = ops.GridSpec(aabb=((-1.5, 1.5),) * 3, density_scale=10.0, density_pre_act=abi.ACT_IDENTITY, density_post_act=abi.ACT_SOFTPLUS)
    params = ops.RenderParams(num_samples=S, near=1.8, far=6.6, perturb=True, white_bkgd=True, image_width=hw)
    ro, rd = ops.cast_rays(hw, hw, float(focal), pose.rotation, pose.translation, dev)
    td, tf = torch.from_numpy(dens).to(dev).requires_grad_(True), torch.from_numpy(feat).to(dev).requires_grad_(True)
    col, dep, acc, _ = ops.render(spec, params, td, tf, ro, rd, rng=(7, 11))
    (col * torch.from_numpy(gc).to(dev)).sum().backward()
    np.testing.assert_array_equal(c_col, col.detach().cpu().numpy().reshape(-1))
    np.testing.assert_array_equal(c_dep, dep.detach().cpu().numpy().reshape(-1))
    np.testing.assert_array_equal(c_acc, acc.detach().cpu().numpy().reshape(-1))
    for got, want in ((c_dd, td.grad), (c_df, tf.grad)):
        want = want.cpu().numpy().reshape(-1)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-6
