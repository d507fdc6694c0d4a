"""GPU parity of the refinement-stage grid passes (graph construction, minimum cut, connected components):
libvoxe_hip.so through the C ABI vs the CPU oracle, bit for bit, plus full-size (160^3) property checks."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from synth import refine_scene
from test_oracle_refine import OFFSETS, pooled_inputs, random_graph

from oracle import voxe_oracle as vo

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import gpu_helpers as gh
    from voxe_hip import ops


def hip_graph_build(dens, feat, sigma, dilate):
    node, cap = ops.graph_build(gh.t(dens), gh.t(feat), sigma=sigma, dilate_yz=dilate)
    return gh.n(node), gh.n(cap)


def hip_graphcut(node, term, cap):
    seg, flow = ops.graphcut(gh.t(node), gh.t(term), gh.t(cap))
    return gh.n(seg), flow


def cut_value(seg, cap):
    """sum of the ORIGINAL capacities that cross from the edit side (0) to the object side (1)"""
    total = 0
    dims = seg.shape
    for d, off in enumerate(OFFSETS):
        sl_a = tuple(slice(max(0, -o), dims[i] - max(0, o)) for i, o in enumerate(off))
        sl_b = tuple(slice(max(0, o), dims[i] - max(0, -o)) for i, o in enumerate(off))
        cross = (seg[sl_a] == 0) & (seg[sl_b] == 1)
        total += int(cap[d][sl_a][cross].astype(np.int64).sum())
    return total


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_graph_build_golden_inputs(tag):
    z = load_golden("refine_graph.npz")
    dens, feat, dilate = pooled_inputs(z, tag)
    sigma = float(z[f"{tag}_kw_sigma"])
    node_o, cap_o = vo.graph_build(dens[..., 0], feat, sigma, dilate)
    node_h, cap_h = hip_graph_build(dens[..., 0], feat, sigma, dilate)
    assert np.array_equal(node_h, node_o)
    assert np.array_equal(cap_h, cap_o)


@pytest.mark.parametrize("dims,F,dilate", [((32, 32, 32), 3, True), ((20, 33, 17), 3, True), ((24, 24, 24), 12, False),
                                           ((1, 9, 40), 3, True), ((48, 40, 2), 27, True)])
def test_graph_build_random(dims, F, dilate):
    rng = np.random.default_rng(sum(dims) + F)
    dens = rng.uniform(-1, 0.4, dims).astype(np.float32)
    dens[rng.uniform(size=dims) < 0.05] = 0.0                     # exact zeros sit on the `> 0` edge
    feat = rng.uniform(0, 1, dims + (F,)).astype(np.float32)
    node_o, cap_o = vo.graph_build(dens, feat, 0.1, dilate)
    node_h, cap_h = hip_graph_build(dens, feat, 0.1, dilate)
    assert np.array_equal(node_h, node_o)
    assert np.array_equal(cap_h, cap_o)
    for d in (0, 2, 4):                                           # n-links are symmetric
        a = cap_h[d][tuple(slice(0, dims[i] - OFFSETS[d][i]) for i in range(3))]
        b = cap_h[d ^ 1][tuple(slice(OFFSETS[d][i], dims[i]) for i in range(3))]
        assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,dims,cap_hi", [(0, (6, 7, 8), 6), (1, (16, 16, 16), 1 << 20), (2, (1, 30, 30), 50),
                                              (3, (32, 24, 40), 1 << 28), (4, (48, 48, 48), 1000), (5, (3, 2, 2), 3)])
def test_graphcut_random_graphs(seed, dims, cap_hi):
    rng = np.random.default_rng(200 + seed)
    node, term, cap = random_graph(rng, dims, 0.75, 1 + int(np.prod(dims)) // 400, 1 + int(np.prod(dims)) // 300, cap_hi)
    seg_o, flow_o, _ = vo.graphcut(node, term, cap)
    seg_h, flow_h = hip_graphcut(node, term, cap)
    assert flow_h == flow_o
    assert np.array_equal(seg_h, seg_o)


def test_graphcut_degenerate():
    node = np.ones((2, 2, 2), np.uint8)
    cap = np.zeros((6, 2, 2, 2), np.int32)
    term = np.zeros((2, 2, 2), np.int8)
    seg, flow = hip_graphcut(node, term, cap)
    assert flow == 0 and (seg == 0).all()
    term[0, 0, 0], term[1, 1, 1] = 1, -1
    cap[:] = 3
    seg, flow = hip_graphcut(node, term, cap)
    assert flow == 9 and seg[0, 0, 0] == 0 and seg[1, 1, 1] == 1
    seg, flow = hip_graphcut(np.zeros((2, 2, 2), np.uint8), term, cap)
    assert flow == 0 and (seg == 255).all()
    # adjacent source / sink seeds only
    node = np.ones((1, 1, 2), np.uint8)
    term = np.array([[[1, -1]]], np.int8)
    cap = np.zeros((6, 1, 1, 2), np.int32)
    cap[4, 0, 0, 0] = cap[5, 0, 0, 1] = 7
    seg, flow = hip_graphcut(node, term, cap)
    assert flow == 7 and seg.tolist() == [[[0, 1]]]


@pytest.mark.parametrize("side", [48, 160])
def test_graphcut_scene_vs_oracle_and_duality(side):
    """the refinement scene at the BASELINE grid size: labels and flow equal the oracle's, and the capacity
    crossing the label boundary equals the flow (max-flow / min-cut duality, independent of any solver)"""
    dens, col, edit, pick = refine_scene(side, n_obj=5000 if side == 160 else 400)
    node_h, cap_h = hip_graph_build(dens[..., 0].numpy(), col.numpy(), 0.1, True)
    node_o, cap_o = vo.graph_build(dens[..., 0].numpy(), col.numpy(), 0.1, True)
    assert np.array_equal(node_h, node_o) and np.array_equal(cap_h, cap_o)
    term = np.zeros((side,) * 3, np.int8)
    term[edit & (node_h > 0)] = 1
    term[tuple(pick.T)] = -1
    seg_h, flow_h = hip_graphcut(node_h, term, cap_h)
    assert cut_value(seg_h, cap_h) == flow_h
    seg_o, flow_o, _ = vo.graphcut(node_o, term, cap_o)
    assert flow_h == flow_o
    assert np.array_equal(seg_h, seg_o)
    assert ((seg_h == 0) | (seg_h == 1)).sum() == (node_h > 0).sum()


@pytest.mark.parametrize("seed,dims,p", [(0, (10, 11, 12), 0.2), (1, (64, 64, 64), 0.08), (2, (5, 130, 7), 0.35),
                                          (3, (96, 80, 72), 0.12), (4, (1, 1, 9), 0.5), (5, (40, 40, 40), 0.6)])
def test_cc_largest_k_random(seed, dims, p):
    rng = np.random.default_rng(seed)
    mask = rng.uniform(size=dims) < p
    for k in (1, 10, 64):
        lab_o, n_o = vo.cc_largest_k(mask, k)
        lab_h, n_h = ops.cc_largest_k(gh.t(mask), k)
        assert n_h == n_o
        assert np.array_equal(gh.n(lab_h), lab_o)


def test_cc_largest_k_scene_160():
    """density > 0 of an edited field at 160^3: one big body + scattered floaters (the post_process_scc input)"""
    dens, _, _, _ = refine_scene(160)
    rng = np.random.default_rng(11)
    mask = dens[..., 0].numpy() > 0
    mask |= rng.uniform(size=mask.shape) < 0.01          # floaters
    mask[80:83] = False                                  # a slab that splits the body in two big parts
    lab_o, n_o = vo.cc_largest_k(mask, 10)
    lab_h, n_h = ops.cc_largest_k(gh.t(mask), 10)
    assert n_h == n_o and n_o > 1000
    assert np.array_equal(gh.n(lab_h), lab_o)
    sizes = np.bincount(lab_o.ravel())
    assert len(sizes) == 11 and (np.diff(sizes[1:]) >= 0).all()     # labels ascend with size; 10 = largest
    empty, n_e = ops.cc_largest_k(torch.zeros((4, 5, 6), dtype=torch.bool, device=gh.DEV), 10)
    assert n_e == 0 and not bool(empty.any())
