"""Attention-grid refinement helpers: masked-L1 attention loss, seed selection, voxel graph cut, edit region.

Interface of the reference's thre3d_atom/modules/refinement_functions.py (`calc_loss_on_attn_grid` :42-76,
`build_graph` :182-298, `get_edit_region` :351-406).  The reference builds the voxel graph with python loops over
every node and solves it with PyMaxflow on the CPU; here the graph construction, the minimum cut and the
labelling are HIP kernels (voxe_graph_build / voxe_graphcut, csrc/voxe_refine.hip) -- the python side only picks
the seed voxels (a few torch reductions over the node list, same calls and RNG use as the reference).
"""
from typing import Tuple

import torch
from torch import Tensor

from voxe_hip import ops as _ops

g_neighbor_offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]]


def calc_loss_on_attn_grid(attn_render: Tensor, attn_map: Tensor, token: str = "", global_step: int = 0,
                           log_freq: int = 50, log_wandb: bool = False) -> Tensor:
    """mean |render - map| over the pixels whose rendered attention is positive (i.e. that hit density)."""
    attn_render = attn_render.reshape(attn_map.shape)
    mask = (attn_render > 0.0).to(attn_map.dtype)
    return ((attn_render - attn_map).abs() * mask).sum() / mask.sum()


def select_seed_voxels(edit_attn_vals: Tensor, obj_attn_vals: Tensor, edit_mask_thresh: float = 0.992,
                       num_obj_voxels_thresh: int = 5000, min_num_edit_voxels: int = 300,
                       top_k_edit_thresh: int = 300, top_k_obj_thresh: int = 200) -> Tuple[Tensor, Tensor]:
    """Indices (into the node list) of the "edit" (source) and "object" (sink) seeds, refinement_functions.py:225-247.

    edit: nodes whose softmax([edit, object]) edit probability is within `edit_mask_thresh` of the best one;
    object: a random subset (torch.randperm on the CPU generator, as the reference) of the nodes where the object
    probability wins.  With fewer than `min_num_edit_voxels` edit nodes both sets fall back to top-k attention."""
    edit = edit_attn_vals.reshape(-1, 1)
    obj = obj_attn_vals.reshape(-1, 1)
    probs = torch.softmax(torch.cat((edit, obj), dim=-1), dim=-1)
    best_edit = probs[:, 0] >= edit_mask_thresh * probs[:, 0].max()
    edit_idx = best_edit.nonzero().reshape(-1)
    obj_candidates = (probs[:, 1] > probs[:, 0]).nonzero().reshape(-1)
    perm = torch.randperm(obj_candidates.shape[0])[:num_obj_voxels_thresh]
    obj_idx = obj_candidates[perm.to(obj_candidates.device)]
    if int(best_edit.sum()) < min_num_edit_voxels:
        edit_idx = torch.topk(edit.reshape(-1), top_k_edit_thresh).indices
        obj_idx = torch.topk(obj.reshape(-1), top_k_obj_thresh).indices
    return edit_idx, obj_idx


def build_graph(features: Tensor, densities: Tensor, edit_attn: Tensor, obj_attn: Tensor, K: float = 0.05,
                sigma: float = 0.1, edit_mask_thresh: float = 0.992, num_obj_voxels_thresh: int = 5000,
                min_num_edit_voxels: int = 300, top_k_edit_thresh: int = 300, top_k_obj_thresh: int = 200,
                downsample_grid: bool = False, downsample_factor: int = 4) -> Tuple[Tensor, Tensor]:
    """Segment the occupied voxels into "edit" (0) and "object" (1) by a minimum cut.

    features [X,Y,Z,F] (already sigmoid'ed colours), densities / edit_attn / obj_attn [X,Y,Z,1].
    Returns (segments [n] int64 on the CPU, voxel indices [n,3] int64 on the CPU) over the graph nodes in memory
    order, like the reference.  `K` multiplies every n-link alike and, with infinite seed links only, cannot
    change the cut (include/voxe.h); it is accepted for interface parity."""
    del K
    if downsample_grid:
        f = int(downsample_factor)
        pool = torch.nn.functional
        density_grid = pool.max_pool3d(densities.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
        feature_grid = pool.avg_pool3d(features.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
        edit_grid = pool.max_pool3d(edit_attn.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
        obj_grid = pool.max_pool3d(obj_attn.permute(3, 0, 1, 2), f, f).permute(1, 2, 3, 0)
    else:
        density_grid, feature_grid, edit_grid, obj_grid = densities, features, edit_attn, obj_attn
    node_mask, cap = _ops.graph_build(density_grid.contiguous(), feature_grid.contiguous(), sigma=sigma,
                                      dilate_yz=not downsample_grid)
    node_lin = node_mask.reshape(-1).nonzero().reshape(-1)
    edit_idx, obj_idx = select_seed_voxels(
        edit_grid.reshape(-1)[node_lin], obj_grid.reshape(-1)[node_lin], edit_mask_thresh, num_obj_voxels_thresh,
        min_num_edit_voxels, top_k_edit_thresh, top_k_obj_thresh)
    terminal = torch.zeros(node_mask.numel(), dtype=torch.int8, device=node_mask.device)
    terminal[node_lin[obj_idx]] = -1
    terminal[node_lin[edit_idx]] = 1          # `if edit ... elif object`: an edit seed wins (:252-255)
    segment, _ = _ops.graphcut(node_mask, terminal.reshape(node_mask.shape), cap)
    segments = segment.reshape(-1)[node_lin].to(torch.int64).cpu()
    _, Y, Z = node_mask.shape
    lin = node_lin.cpu()
    idx_values = torch.stack((lin // (Y * Z), (lin // Z) % Y, lin % Z), dim=-1)
    return segments, idx_values


def get_edit_region(vol_mod_edit, vol_mod_object, vol_mod_output, downsample_grid: bool = False,
                    downsample_factor: int = 4, K: float = 5.0, sigma: float = 0.1, edit_mask_thresh: float = 0.992,
                    num_obj_voxels_thresh: int = 5000, min_num_edit_voxels: int = 300, top_k_edit_thresh: int = 300,
                    top_k_obj_thresh: int = 200) -> None:
    """Graph-cut the two optimised attention grids into an edit region and store it as the output model's attention
    grid: 0 on edit voxels, -5 on other occupied voxels, -10 elsewhere (refinement_functions.py:351-406)."""
    edit_repr, obj_repr = vol_mod_edit.thre3d_repr, vol_mod_object.thre3d_repr
    if not torch.equal(edit_repr._densities, obj_repr._densities):
        raise AssertionError("ERROR: Density values for edit and object grids don't match")
    if not torch.equal(edit_repr._features, obj_repr._features):
        raise AssertionError("ERROR: Feature values for edit and object grids don't match")
    with torch.no_grad():
        densities = edit_repr._densities.detach()
        edit_attn, obj_attn = edit_repr.attn.detach(), obj_repr.attn.detach()
        colours = torch.sigmoid(edit_repr._features.detach())
        ids, idxs = build_graph(colours, densities, edit_attn, obj_attn, K=K, sigma=sigma,
                                edit_mask_thresh=edit_mask_thresh, num_obj_voxels_thresh=num_obj_voxels_thresh,
                                min_num_edit_voxels=min_num_edit_voxels, top_k_edit_thresh=top_k_edit_thresh,
                                top_k_obj_thresh=top_k_obj_thresh, downsample_grid=downsample_grid,
                                downsample_factor=downsample_factor)
        keep_grid = torch.full_like(edit_attn, -10.0)
        keep_grid[densities > 0.0] = -5.0
        factor = int(downsample_factor) if downsample_grid else 1
        X, Y, Z = (int(v) for v in densities.shape[:3])
        coarse = torch.zeros((-(-X // factor), -(-Y // factor), -(-Z // factor)), dtype=torch.bool, device=densities.device)
        e = idxs[ids == 0].to(densities.device)
        coarse[e[:, 0], e[:, 1], e[:, 2]] = True
        fine = coarse
        if factor > 1:  # every coarse edit voxel covers a factor^3 block of the full grid (:399-402)
            for axis in range(3):
                fine = fine.repeat_interleave(factor, dim=axis)
            fine = fine[:X, :Y, :Z]
        keep_grid[fine] = 0.0
        vol_mod_output.thre3d_repr.attn = torch.nn.Parameter(keep_grid)


def restore_outside_largest_component(vol_mod, pretrained_vol_mod, k: int = 10) -> int:
    """`post_process_scc` of the reference's edit script (edit_pretrained_relu_field.py:381-391,408-418): label the
    26-connected components of `density > 0`, keep the k largest numbered 1..k by ascending size, and give every
    voxel whose label is not k (i.e. everything outside the single largest component) the density of the
    un-edited field.  Returns the number of components.  Like the reference, fewer than k components leave no
    label k, so every density is restored."""
    grid, ref = vol_mod.thre3d_repr, pretrained_vol_mod.thre3d_repr
    with torch.no_grad():
        dens = grid._densities.detach()
        labels, num = _ops.cc_largest_k((dens > 0).squeeze(-1), k)
        outside = (labels != k).unsqueeze(-1)
        new = torch.where(outside, ref._densities.detach().to(dens.device), dens)
    grid._densities = torch.nn.Parameter(new)
    return num
