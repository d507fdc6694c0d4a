"""Shared helpers of the parity tests (oracle-side: numpy only)."""
import numpy as np

from voxe_hip import abi
from voxe_hip.desc import make_render_cfg

from oracle import voxe_oracle as vo

KINDS = {
    # kind -> (density_pre_act, density_post_act, expected_density_scale) as built by tools/gen_golden.py
    "softplus": (abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, 100.0 / 3.0),
    "softplus_soft": (abi.ACT_IDENTITY, abi.ACT_SOFTPLUS, 2.0),
    "relu": (abi.ACT_IDENTITY, abi.ACT_RELU, 100.0 / 3.0),
    "abs": (abi.ACT_ABS, abi.ACT_IDENTITY, 1.0),
}


def grid_from_golden(g, prefix, kind, attn=False):
    pre, post, scale = KINDS[kind]
    aabb = [tuple(r) for r in g[prefix + "aabb"]]
    feats = g[prefix + ("attn" if attn else "features")]
    return vo.Grid(g[prefix + "densities"], feats, aabb, scale, pre, post,
                   abi.FEAT_ATTN if attn else abi.FEAT_SH)


def cfg_from_bounds(bounds, S, **kw):
    return make_render_cfg(S, float(bounds[0]), float(bounds[1]), **kw)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * np.log10(mse)


def nan_equal(a, b, rtol, atol):
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "NaN pattern differs"
    np.testing.assert_allclose(a[~na], b[~nb], rtol=rtol, atol=atol)


def band_errors(got, ref, bands):
    """relative error of `got` per magnitude band of the reference: for every (lo, hi] in `bands` (fractions of max |ref|)
    that holds at least one element -> (count, median of |got - ref| / |ref|, 99th percentile)"""
    ref = np.asarray(ref, np.float64).ravel()
    err = np.abs(np.asarray(got, np.float64).ravel() - ref)
    mag = np.abs(ref)
    big = float(mag.max())
    out = {}
    for lo, hi in bands:
        m = (mag > lo * big) & (mag <= hi * big)
        if m.any():
            rel = err[m] / mag[m]
            out[(lo, hi)] = (int(m.sum()), float(np.median(rel)), float(np.quantile(rel, 0.99)))
    return out
