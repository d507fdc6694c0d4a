"""numpy front-end of the CPU oracle (oracle/libvoxe_oracle.so, built from oracle/voxe_cpu.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (vox-e_amd/) must never import this module.

Parity status: PINNED against tests/golden/*.npz (vectors produced by importing the reference in
the build container with tools/gen_golden.py); see tests/test_oracle_vs_golden.py.
"""
import ctypes as C
import os
import subprocess
import sys
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_PKG = os.path.join(_ROOT, "vox-e_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from voxe_hip import abi  # noqa: E402  (ctypes mirror of include/voxe.h -- the shared ABI description)
from voxe_hip.desc import make_grid_desc, make_render_cfg  # noqa: E402

_LIB_PATH = os.path.join(_HERE, "libvoxe_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    deps = [os.path.join(_HERE, "voxe_cpu.c"), os.path.join(_HERE, "voxe_cpu_refine.c"),
            os.path.join(_ROOT, "include", "voxe.h")]
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps)
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvoxe_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = abi.declare(C.CDLL(_LIB_PATH), "voxe_cpu_")
    return _lib


def _check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"oracle {what} failed with status {status}")


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


@dataclass
class Grid:
    densities: np.ndarray  # [X,Y,Z,1]
    features: np.ndarray  # [X,Y,Z,F]
    aabb: Sequence[Tuple[float, float]]
    density_scale: float = 1.0
    density_pre_act: int = abi.ACT_IDENTITY
    density_post_act: int = abi.ACT_SOFTPLUS
    feature_kind: int = abi.FEAT_SH

    def __post_init__(self):
        self.densities = _f32(self.densities)
        self.features = _f32(self.features)

    def desc(self) -> abi.VoxeGridDesc:
        X, Y, Z, F = self.features.shape
        return make_grid_desc(
            self.densities.ctypes.data, self.features.ctypes.data, (X, Y, Z), F, self.aabb,
            self.density_scale, self.density_pre_act, self.density_post_act, self.feature_kind,
        )

    @property
    def cout(self) -> int:
        return 1 if self.feature_kind == abi.FEAT_ATTN else 3


def aabb_from_voxel_size(dims, voxel_size, location=(0.0, 0.0, 0.0)):
    """VoxelGrid._setup_bounding_box_planes, thre3d_atom/thre3d_reprs/voxels.py:196-223 (python floats)."""
    out = []
    for n, vs, c in zip(dims, voxel_size, location):
        half = (n * vs) / 2
        out.append((c - half, c + half))
    return out


def num_threads() -> int:
    return int(lib().voxe_cpu_num_threads())


def jitter_uniform(seed: int, rng_offset: int, ray: int, sample: int) -> float:
    return float(lib().voxe_cpu_jitter_uniform(seed, rng_offset, ray, sample))


def jitter_stream(seed: int, rng_offset: int, R: int, S: int) -> np.ndarray:
    out = np.empty((R, S), np.float32)
    f = lib().voxe_cpu_jitter_uniform
    for r in range(R):
        for k in range(S):
            out[r, k] = f(seed, rng_offset, r, k)
    return out


def cast_rays(H: int, W: int, focal: float, rot, trans):
    rot = _f32(rot).reshape(9)
    trans = _f32(trans).reshape(3)
    o = np.empty((H * W, 3), np.float32)
    d = np.empty((H * W, 3), np.float32)
    fp = C.POINTER(C.c_float)
    _check(
        lib().voxe_cpu_cast_rays(H, W, float(focal), rot.ctypes.data_as(fp), trans.ctypes.data_as(fp),
                                 o.ctypes.data, d.ctypes.data),
        "cast_rays",
    )
    return o, d


def cast_rays_indexed(H: int, W: int, focal: float, poses, flat_index):
    poses = _f32(poses).reshape(-1, 12)
    idx = np.ascontiguousarray(flat_index, dtype=np.int64)
    o = np.empty((idx.shape[0], 3), np.float32)
    d = np.empty((idx.shape[0], 3), np.float32)
    _check(lib().voxe_cpu_cast_rays_indexed(H, W, float(focal), poses.ctypes.data, poses.shape[0], idx.ctypes.data,
                                            idx.shape[0], o.ctypes.data, d.ctypes.data), "cast_rays_indexed")
    return o, d


def random_subset(n: int, count: int, seed: int, rng_offset: int) -> np.ndarray:
    out = np.empty((count,), np.int64)
    _check(lib().voxe_cpu_random_subset(int(n), int(count), int(seed), int(rng_offset), out.ctypes.data), "random_subset")
    return out


def render_fwd(grid: Grid, cfg: abi.VoxeRenderCfg, rays_o, rays_d, jitter=None):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    R = rays_o.shape[0]
    jitter = None if jitter is None else _f32(jitter)
    colour = np.empty((R, grid.cout), np.float32)
    depth = np.empty((R,), np.float32)
    acc = np.empty((R,), np.float32)
    disp = np.empty((R,), np.float32)
    g = grid.desc()
    _check(
        lib().voxe_cpu_render_fwd(C.byref(g), C.byref(cfg), rays_o.ctypes.data, rays_d.ctypes.data, R,
                                  _ptr(jitter), colour.ctypes.data, depth.ctypes.data, acc.ctypes.data,
                                  disp.ctypes.data),
        "render_fwd",
    )
    return {"colour": colour, "depth": depth, "acc": acc, "disparity": disp}


def render_bwd(grid: Grid, cfg: abi.VoxeRenderCfg, rays_o, rays_d, d_colour, d_depth=None, d_acc=None,
               jitter=None, want_densities=True, want_features=True):
    rays_o, rays_d, d_colour = _f32(rays_o), _f32(rays_d), _f32(d_colour)
    R = rays_o.shape[0]
    jitter = None if jitter is None else _f32(jitter)
    d_depth = None if d_depth is None else _f32(d_depth)
    d_acc = None if d_acc is None else _f32(d_acc)
    gd = np.zeros_like(grid.densities) if want_densities else None
    gf = np.zeros_like(grid.features) if want_features else None
    g = grid.desc()
    _check(
        lib().voxe_cpu_render_bwd(C.byref(g), C.byref(cfg), rays_o.ctypes.data, rays_d.ctypes.data, R,
                                  _ptr(jitter), d_colour.ctypes.data, _ptr(d_depth), _ptr(d_acc),
                                  _ptr(gd), _ptr(gf), 0),
        "render_bwd",
    )
    return gd, gf


def sample_probe(grid: Grid, cfg: abi.VoxeRenderCfg, rays_o, rays_d, jitter=None):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    R, S = rays_o.shape[0], cfg.num_samples
    jitter = None if jitter is None else _f32(jitter)
    idx = np.empty((R, S, 3), np.int32)
    inside = np.empty((R, S), np.uint8)
    z = np.empty((R, S), np.float32)
    sigma = np.empty((R, S), np.float32)
    rad = np.empty((R, S, grid.cout), np.float32)
    g = grid.desc()
    _check(
        lib().voxe_cpu_sample_probe(C.byref(g), C.byref(cfg), rays_o.ctypes.data, rays_d.ctypes.data, R,
                                    _ptr(jitter), idx.ctypes.data, inside.ctypes.data, z.ctypes.data,
                                    sigma.ctypes.data, rad.ctypes.data),
        "sample_probe",
    )
    return {"idx": idx, "inside": inside.astype(bool), "z": z, "sigma": sigma, "rad": rad}


def query_fwd(grid: Grid, points):
    points = _f32(points)
    N, F = points.shape[0], grid.features.shape[-1]
    out = np.empty((N, F + 1), np.float32)
    g = grid.desc()
    _check(lib().voxe_cpu_query_fwd(C.byref(g), points.ctypes.data, N, out.ctypes.data), "query_fwd")
    return out


def query_bwd(grid: Grid, points, d_out):
    points, d_out = _f32(points), _f32(d_out)
    gd, gf = np.zeros_like(grid.densities), np.zeros_like(grid.features)
    g = grid.desc()
    _check(lib().voxe_cpu_query_bwd(C.byref(g), points.ctypes.data, points.shape[0], d_out.ctypes.data,
                                    gd.ctypes.data, gf.ctypes.data, 0), "query_bwd")
    return gd, gf


def dcl_fwd_bwd(a, b, grad_scale: float = 1.0):
    a, b = _f32(a), _f32(b)
    loss = np.empty((1,), np.float32)
    d_a = np.empty_like(a)
    _check(lib().voxe_cpu_dcl_fwd_bwd(a.ctypes.data, b.ctypes.data, a.size, float(grad_scale),
                                      loss.ctypes.data, d_a.ctypes.data, 0), "dcl")
    return float(loss[0]), d_a


def density_diff_fwd_bwd(a, b, kind: int, grad_scale: float = 1.0):
    """l2_mode (kind = abi.DREG_L2) / l1_mode (abi.DREG_L1) of density_correlation_loss_fn: (loss, gradient w.r.t. a)"""
    a, b = _f32(a), _f32(b)
    loss = np.empty((1,), np.float32)
    d_a = np.empty_like(a)
    _check(lib().voxe_cpu_density_diff_fwd_bwd(a.ctypes.data, b.ctypes.data, a.size, int(kind), float(grad_scale),
                                               loss.ctypes.data, d_a.ctypes.data, 0), "density_diff")
    return float(loss[0]), d_a


def feature_correlation_fwd_bwd(f, r, grad_scale: float = 1.0):
    """_feature_correlation_loss: (loss, gradient w.r.t. f); f, r [..., F]"""
    f, r = _f32(f), _f32(r)
    loss = np.empty((1,), np.float32)
    d_f = np.empty_like(f)
    F = f.shape[-1]
    _check(lib().voxe_cpu_feature_correlation_fwd_bwd(f.ctypes.data, r.ctypes.data, f.size // F, F, float(grad_scale),
                                                      loss.ctypes.data, d_f.ctypes.data, 0), "feature_correlation")
    return float(loss[0]), d_f


def tv_fwd_bwd(grid, grad_scale: float = 1.0):
    grid = _f32(grid)
    X, Y, Z, Cn = grid.shape
    loss = np.empty((1,), np.float32)
    d_g = np.empty_like(grid)
    _check(lib().voxe_cpu_tv_fwd_bwd(grid.ctypes.data, X, Y, Z, Cn, float(grad_scale), loss.ctypes.data,
                                     d_g.ctypes.data, 0), "tv")
    return float(loss[0]), d_g


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """In place on float32 contiguous numpy arrays."""
    for a in (param, grad, exp_avg, exp_avg_sq):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    _check(lib().voxe_cpu_adam_step(param.ctypes.data, grad.ctypes.data, exp_avg.ctypes.data,
                                    exp_avg_sq.ctypes.data, param.size, lr, beta1, beta2, eps, step), "adam")


def upsample_trilinear(src, out_size):
    src = _f32(src)
    X, Y, Z, Cn = src.shape
    X2, Y2, Z2 = out_size
    dst = np.empty((X2, Y2, Z2, Cn), np.float32)
    _check(lib().voxe_cpu_upsample_trilinear(src.ctypes.data, X, Y, Z, Cn, dst.ctypes.data, X2, Y2, Z2),
           "upsample")
    return dst


# ---- refinement stage -------------------------------------------------------------------------------------
def graph_build(density_grid, feature_grid, sigma: float = 0.1, dilate_yz: bool = True):
    """-> node_mask u8 [X,Y,Z], cap int32 [6,X,Y,Z]"""
    dens = _f32(density_grid).reshape(np.asarray(density_grid).shape[:3])
    feat = _f32(feature_grid)
    X, Y, Z = dens.shape
    F = feat.shape[-1]
    node = np.empty((X, Y, Z), np.uint8)
    cap = np.empty((6, X, Y, Z), np.int32)
    _check(lib().voxe_cpu_graph_build(dens.ctypes.data, feat.ctypes.data, X, Y, Z, F, float(sigma),
                                      int(bool(dilate_yz)), node.ctypes.data, cap.ctypes.data), "graph_build")
    return node, cap


def graphcut(node_mask, terminal, cap):
    """-> segment u8 [X,Y,Z] (255 = no node), flow (int), residual capacities"""
    node = np.ascontiguousarray(node_mask, dtype=np.uint8)
    term = np.ascontiguousarray(terminal, dtype=np.int8)
    res = np.array(cap, dtype=np.int32, order="C", copy=True)
    X, Y, Z = node.shape
    seg = np.empty((X, Y, Z), np.uint8)
    flow = np.zeros((1,), np.int64)
    _check(lib().voxe_cpu_graphcut(node.ctypes.data, term.ctypes.data, res.ctypes.data, X, Y, Z,
                                   seg.ctypes.data, flow.ctypes.data), "graphcut")
    return seg, int(flow[0]), res


def cc_largest_k(mask, k: int):
    """-> labels int32 [X,Y,Z], number of 26-connected components"""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    X, Y, Z = m.shape
    labels = np.empty((X, Y, Z), np.int32)
    n = np.zeros((1,), np.int32)
    _check(lib().voxe_cpu_cc_largest_k(m.ctypes.data, X, Y, Z, int(k), labels.ctypes.data, n.ctypes.data),
           "cc_largest_k")
    return labels, int(n[0])
