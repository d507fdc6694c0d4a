"""Pins the refinement-stage oracle (oracle/voxe_cpu_refine.c) -- CPU only.

* graph construction  vs the node / n-link lists the reference's build_graph emitted (tests/golden/refine_graph.npz)
* minimum cut         vs scipy.sparse.csgraph.maximum_flow and vs brute-force enumeration of every cut
* connected components vs scipy.ndimage.label (full 3x3x3 connectivity)
"""
import itertools

import numpy as np
import pytest
import torch

from conftest import load_golden as golden

from oracle import voxe_oracle as vo

ONE = float(1 << 28)
OFFSETS = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])


def pooled_inputs(z, tag):
    """inputs of the graph construction: the grids themselves, or the pooled ones of the down-sampled branch
    (refinement_functions.py:189-196: max-pooled densities, average-pooled features)"""
    dens = torch.from_numpy(z[f"{tag}_densities"])
    feat = torch.from_numpy(z[f"{tag}_features"])
    if f"{tag}_kw_downsample_grid" in z.files and bool(z[f"{tag}_kw_downsample_grid"]):
        f = int(z[f"{tag}_kw_downsample_factor"])
        dens = torch.nn.functional.max_pool3d(dens.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
        feat = torch.nn.functional.avg_pool3d(feat.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
        return dens.contiguous().numpy(), feat.contiguous().numpy(), False
    return dens.numpy(), feat.numpy(), True


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_graph_build_matches_reference_graph(tag):
    z = golden("refine_graph.npz")
    dens, feat, dilate = pooled_inputs(z, tag)
    K, sigma = float(z[f"{tag}_kw_K"]), float(z[f"{tag}_kw_sigma"])
    node, cap = vo.graph_build(dens[..., 0], feat, sigma=sigma, dilate_yz=dilate)
    idx = z[f"{tag}_node_idx"]
    assert np.array_equal(np.argwhere(node > 0), idx)          # same nodes, same (raster) order
    # accumulate the reference's add_edge(i, j, w, w) calls into directed capacities
    want = np.zeros(cap.shape, np.float64)
    for i, j, w, rw in z[f"{tag}_edges"]:
        a, b = idx[int(i)], idx[int(j)]
        d = int(np.flatnonzero((OFFSETS == (b - a)).all(1))[0])
        want[d, a[0], a[1], a[2]] += w
        want[d ^ 1, b[0], b[1], b[2]] += rw
    got = cap.astype(np.float64) * (K / ONE)
    assert np.array_equal(want > 0, cap > 0)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=K * 4 / ONE)


def brute_force_sink_side(node, term, cap):
    """minimum cut value and the intersection of the sink sides of ALL minimum cuts (= the minimal sink side)"""
    free = [tuple(p) for p in np.argwhere((node > 0) & (term == 0))]
    assert len(free) <= 16
    nodes = [tuple(p) for p in np.argwhere(node > 0)]
    best, inter = None, None
    for bits in itertools.product((0, 1), repeat=len(free)):
        side = {p: (0 if term[p] > 0 else 1) for p in nodes if term[p] != 0}
        side.update(dict(zip(free, bits)))
        val = 0
        for p in nodes:
            if side[p] != 0:
                continue
            for d, off in enumerate(OFFSETS):
                q = tuple(np.add(p, off))
                if q in side and side[q] == 1:
                    val += int(cap[(d,) + p])
        sink = {p for p in nodes if side[p] == 1}
        if best is None or val < best:
            best, inter = val, sink
        elif val == best:
            inter &= sink
    return best, inter


def random_graph(rng, dims, p_node, n_src, n_snk, cap_hi):
    node = (rng.uniform(size=dims) < p_node).astype(np.uint8)
    cap = np.zeros((6,) + dims, np.int32)
    for p in np.argwhere(node > 0):
        for d in (0, 2, 4):
            q = p + OFFSETS[d]
            if (q < dims).all() and node[tuple(q)] and rng.uniform() < 0.8:
                c = int(rng.integers(1, cap_hi))
                cap[(d,) + tuple(p)] = c
                cap[(d ^ 1,) + tuple(q)] = c if rng.uniform() < 0.7 else int(rng.integers(0, cap_hi))
    term = np.zeros(dims, np.int8)
    cells = rng.permutation(np.argwhere(node > 0))
    for p in cells[:n_src]:
        term[tuple(p)] = 1
    for p in cells[n_src:n_src + n_snk]:
        term[tuple(p)] = -1
    return node, term, cap


@pytest.mark.parametrize("seed", range(12))
def test_graphcut_vs_brute_force(seed):
    rng = np.random.default_rng(seed)
    dims = (2, 3, 3) if seed % 2 else (3, 2, 2)
    node, term, cap = random_graph(rng, dims, 0.9, 2, 2, 6 if seed < 6 else 3)  # small capacities => many ties
    seg, flow, _ = vo.graphcut(node, term, cap)
    best, sink = brute_force_sink_side(node, term, cap)
    assert flow == best
    assert {tuple(p) for p in np.argwhere(seg == 1)} == sink
    assert np.array_equal(seg == 255, node == 0)


def scipy_cut(node, term, cap):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order, maximum_flow

    dims = node.shape
    ids = -np.ones(dims, np.int64)
    pts = np.argwhere(node > 0)
    ids[tuple(pts.T)] = np.arange(len(pts))
    n = len(pts)
    S, T = n, n + 1
    big = int(cap.astype(np.int64).sum()) + 1
    rows, cols, vals = [], [], []
    for i, p in enumerate(pts):
        for d, off in enumerate(OFFSETS):
            q = p + off
            if (q >= 0).all() and (q < dims).all() and node[tuple(q)] and cap[(d,) + tuple(p)] > 0:
                rows.append(i), cols.append(int(ids[tuple(q)])), vals.append(int(cap[(d,) + tuple(p)]))
        if term[tuple(p)] > 0:
            rows.append(S), cols.append(i), vals.append(big)
        elif term[tuple(p)] < 0:
            rows.append(i), cols.append(T), vals.append(big)
    g = csr_matrix((np.array(vals, np.int64), (rows, cols)), shape=(n + 2, n + 2))
    res = maximum_flow(g, S, T)
    residual = (g - res.flow).tocsr()          # flow is antisymmetric: reverse arcs gain what forward arcs lose
    residual.data = np.maximum(residual.data, 0)
    residual.eliminate_zeros()
    reach = breadth_first_order(residual.T.tocsr(), T, directed=True, return_predecessors=False)
    sink_side = np.zeros(n + 2, bool)
    sink_side[reach] = True
    seg = np.full(dims, 255, np.uint8)
    seg[tuple(pts.T)] = sink_side[:n].astype(np.uint8)
    return seg, int(res.flow_value)


@pytest.mark.parametrize("seed,dims", [(0, (6, 7, 8)), (1, (12, 9, 10)), (2, (16, 16, 16)), (3, (1, 20, 20)),
                                        (4, (24, 20, 18))])
def test_graphcut_vs_scipy_max_flow(seed, dims):
    rng = np.random.default_rng(100 + seed)
    node, term, cap = random_graph(rng, dims, 0.7, 5, 7, 1 << 20)
    seg, flow, res = vo.graphcut(node, term, cap)
    seg_ref, flow_ref = scipy_cut(node, term, cap)
    assert flow == flow_ref
    assert np.array_equal(seg, seg_ref)
    assert (res >= 0).all()


def test_graphcut_on_reference_shaped_graph():
    """capacities from the graph builder (with exact ties: multiplicity 1 and 2 of the same quantum)"""
    z = golden("refine_graph.npz")
    dens, feat, dilate = pooled_inputs(z, "a")
    node, cap = vo.graph_build(dens[..., 0], feat, sigma=0.1, dilate_yz=dilate)
    idx = z["a_node_idx"]
    term = np.zeros(node.shape, np.int8)
    for i, s, t in z["a_tedges"]:
        term[tuple(idx[int(i)])] = 1 if np.isinf(s) else -1
    seg, flow, _ = vo.graphcut(node, term, cap)
    seg_ref, flow_ref = scipy_cut(node, term, cap)
    assert flow == flow_ref and np.array_equal(seg, seg_ref)
    assert (seg == 0).sum() >= (term > 0).sum() and (seg == 1).sum() >= (term < 0).sum()


def test_graphcut_degenerate_inputs():
    node = np.ones((2, 2, 2), np.uint8)
    cap = np.zeros((6, 2, 2, 2), np.int32)
    term = np.zeros((2, 2, 2), np.int8)
    seg, flow, _ = vo.graphcut(node, term, cap)            # no seeds, no edges: everything is "edit" (default SOURCE)
    assert flow == 0 and (seg == 0).all()
    term[0, 0, 0], term[1, 1, 1] = 1, -1
    cap[:] = 3
    seg, flow, _ = vo.graphcut(node, term, cap)
    assert flow == 9 and seg[0, 0, 0] == 0 and seg[1, 1, 1] == 1
    seg2, flow2, _ = vo.graphcut(np.zeros((2, 2, 2), np.uint8), term, cap)   # no nodes at all
    assert flow2 == 0 and (seg2 == 255).all()


@pytest.mark.parametrize("seed,dims,p", [(0, (10, 11, 12), 0.2), (1, (16, 16, 16), 0.1), (2, (5, 30, 7), 0.35),
                                          (3, (20, 20, 20), 0.05), (4, (1, 1, 9), 0.5)])
def test_cc_largest_k_vs_scipy(seed, dims, p):
    import scipy.ndimage as ndi

    rng = np.random.default_rng(seed)
    mask = rng.uniform(size=dims) < p
    lab, n = ndi.label(mask, structure=np.ones((3, 3, 3)))
    for k in (1, 3, 10, 10_000):
        labels, ncomp = vo.cc_largest_k(mask, k)
        assert ncomp == n
        sizes = np.bincount(lab.ravel())[1:]
        first = ndi.minimum(np.arange(mask.size).reshape(dims), lab, index=np.arange(1, n + 1)) if n else []
        order = sorted(range(n), key=lambda c: (-sizes[c], first[c]))     # larger first, then earlier first voxel
        M = min(k, n)
        want = np.zeros(dims, np.int32)
        for rank, c in enumerate(order[:M]):
            want[lab == c + 1] = M - rank
        assert np.array_equal(labels, want)


def test_cc_no_foreground():
    labels, n = vo.cc_largest_k(np.zeros((3, 4, 5), bool), 10)
    assert n == 0 and not labels.any()


@pytest.mark.parametrize("n,count", [(1, 1), (2, 2), (7, 5), (1000, 1000), (8 * 400 * 400, 32768), (1 << 20, 4096),
                                     ((1 << 20) + 1, 50000)])
def test_random_subset_is_a_uniform_looking_set_of_distinct_indices(n, count):
    """the Feistel sampler that stands in for torch.randperm(n)[:count]: distinct, in range, different per (seed,
    offset), a full permutation when count == n, and evenly spread (chi-square over 64 buckets)"""
    a = vo.random_subset(n, count, 1234, 1)
    assert a.shape == (count,) and a.min() >= 0 and a.max() < n
    assert len(np.unique(a)) == count
    if count == n:
        assert np.array_equal(np.sort(a), np.arange(n))
    if n > 1000:
        b = vo.random_subset(n, count, 1234, 2)
        assert len(np.intersect1d(a, b)) < count * (count / n) * 3 + 50          # ~ count^2 / n by chance
        assert not np.array_equal(a, vo.random_subset(n, count, 1235, 1))
        assert np.array_equal(a, vo.random_subset(n, count, 1234, 1))             # reproducible
        if count >= 4096:
            hist = np.bincount((a * 64 // n).astype(np.int64), minlength=64).astype(np.float64)
            chi2 = float(((hist - count / 64) ** 2 / (count / 64)).sum())
            assert chi2 < 120.0, chi2                                             # 63 d.o.f.: mean 63, p(>120) ~ 2e-5
            # order is random too: successive picks are not monotone
            assert 0.4 < float((np.diff(a) > 0).mean()) < 0.6
