"""ctypes mirror of include/voxe.h (structs, enums, function prototypes).

No torch import here: this module only describes the C ABI.  `declare(lib, prefix)` installs the
argtypes/restype of every entry point on a loaded library; prefix "voxe_" is the HIP product
library (libvoxe_hip.so), prefix "voxe_cpu_" is the test oracle (oracle/libvoxe_oracle.so), which
shares the same signatures minus (workspace, stream).
"""
import ctypes as C

ABI_VERSION = 12

# VoxeStatus
OK = 0
ERR_NULL_POINTER = -1
ERR_BAD_SHAPE = -2
ERR_UNSUPPORTED = -3
ERR_WORKSPACE = -4
ERR_LAUNCH = -5
ERR_NO_DEVICE = -6

# VoxeAct
ACT_IDENTITY = 0
ACT_ABS = 1
ACT_RELU = 2
ACT_SOFTPLUS = 3

# density regulariser kinds of the edit (VOXE_DREG_*)
DREG_CORRELATION, DREG_L2, DREG_L1 = 0, 1, 2

# VoxeFeatureKind
FEAT_SH = 0
FEAT_ATTN = 1

_f3 = C.c_float * 3


class VoxeGridDesc(C.Structure):
    _fields_ = [
        ("densities", C.c_void_p),
        ("features", C.c_void_p),
        ("X", C.c_int32),
        ("Y", C.c_int32),
        ("Z", C.c_int32),
        ("F", C.c_int32),
        ("aabb_lo", _f3),
        ("aabb_hi", _f3),
        ("norm_scale", _f3),
        ("norm_bias", _f3),
        ("density_scale", C.c_float),
        ("density_pre_act", C.c_int32),
        ("density_post_act", C.c_int32),
        ("feature_kind", C.c_int32),
    ]


class VoxeDispatch(C.Structure):
    """which kernels render a call / with which tuning parameters; every field 0 = the shipped default (include/voxe.h)"""
    _fields_ = [
        ("bwd_mode", C.c_int32),
        ("tile_map", C.c_int32),
        ("tile_min_rays", C.c_int64),
        ("tile_two_phase", C.c_int32),
        ("tile_qsplit", C.c_int32),
        ("tile_kl", C.c_int32),
        ("tile_fit_m", C.c_float),
        ("tile_fit_lat", C.c_float),
        ("fwd_window", C.c_int32),
        ("fwd_fit_lat", C.c_float),
        ("fwd_fit_m", C.c_float),
        ("fwd_zdom", C.c_float),
        ("fwd_max_adv", C.c_float),
        ("fwd_segments_per_thread", C.c_int32),
        ("region_min_rays", C.c_int64),
        ("region_image_ratio", C.c_float),
        ("tile_lean", C.c_int32),
        ("precise_grad", C.c_int32),
        ("region_lds_ranks", C.c_int32),
        ("tile_phases", C.c_int32),
    ]


class VoxeRenderCfg(C.Structure):
    _fields_ = [
        ("num_samples", C.c_int32),
        ("near", C.c_float),
        ("far", C.c_float),
        ("perturb", C.c_int32),
        ("linear_disparity", C.c_int32),
        ("aabb_clip", C.c_int32),
        ("white_bkgd", C.c_int32),
        ("sh_degree", C.c_int32),
        ("render_diffuse", C.c_int32),
        ("term_eps", C.c_float),
        ("seed", C.c_uint64),
        ("rng_offset", C.c_uint64),
        ("reuse_packed_grid", C.c_int32),
        ("image_width", C.c_int32),
        ("image_height", C.c_int32),
        ("deterministic", C.c_int32),
        ("linear_grad", C.c_int32),
        ("ray_state_valid", C.c_int32),
        ("dispatch", C.POINTER(VoxeDispatch)),   # NULL = the shipped dispatch
    ]


class VoxeProfile(C.Structure):
    _fields_ = [
        ("ms_pack", C.c_double), ("ms_fwd", C.c_double), ("ms_memset", C.c_double), ("ms_bwd", C.c_double),
        ("ms_unpack", C.c_double),
        ("n_pack", C.c_int32), ("n_fwd", C.c_int32), ("n_memset", C.c_int32), ("n_bwd", C.c_int32),
        ("n_unpack", C.c_int32), ("n_dropped", C.c_int32),
    ]


class VoxeGridRegularisers(C.Structure):
    _fields_ = [
        ("dcl_reference", C.c_void_p), ("dcl_weight", C.c_float), ("dcl_loss", C.c_void_p),
        ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
        ("density_kind", C.c_int32), ("feat_reference", C.c_void_p), ("feat_weight", C.c_float), ("feat_loss", C.c_void_p),
    ]


class VoxeReconStep(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("focal", C.c_float),
        ("poses", C.c_void_p), ("image_rows", C.c_void_p), ("images", C.c_void_p), ("num_images", C.c_int32),
        ("K", C.c_int32), ("batch", C.c_int64), ("diffuse_regularisation", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("step_densities", C.c_int64), ("step_features", C.c_int64),
        ("exp_avg_densities", C.c_void_p), ("exp_avg_sq_densities", C.c_void_p),
        ("exp_avg_features", C.c_void_p), ("exp_avg_sq_features", C.c_void_p),
        ("losses", C.c_void_p), ("zero_gradient_first", C.c_int32),
    ]


class VoxeAttnRefineStep(C.Structure):
    _fields_ = [
        ("attn_map", C.c_void_p), ("tv_weight", C.c_float), ("tv_loss_always", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("step", C.c_int64), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("losses", C.c_void_p), ("attn_render", C.c_void_p), ("zero_gradient_first", C.c_int32),
    ]


_P = C.c_void_p
_GD = C.POINTER(VoxeGridDesc)
_RC = C.POINTER(VoxeRenderCfg)

# name -> (restype, argtypes, takes_workspace, takes_stream); the oracle twin drops the last two groups
_COMMON = {
    "cast_rays": (C.c_int, [C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P], False, True),
    "cast_rays_indexed": (C.c_int, [C.c_int32, C.c_int32, C.c_float, _P, C.c_int32, _P, C.c_int64, _P, _P], False, True),
    "random_subset": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, _P], False, True),
    "render_fwd": (C.c_int, [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P], True, True),
    "render_bwd": (C.c_int, [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32], True, True),
    "sample_probe": (C.c_int, [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P], True, True),
    "dcl_fwd_bwd": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P, _P, C.c_int32], True, True),
    "density_diff_fwd_bwd": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_float, _P, _P, C.c_int32], True, True),
    "feature_correlation_fwd_bwd": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_float, _P, _P, C.c_int32], True, True),
    "tv_fwd_bwd": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, C.c_int32], True, True),
    "adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64], False, True),
    "upsample_trilinear": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32], False, True),
    # refinement stage (graph cut / connected components)
    "graph_build": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, _P], False, True),
    "graphcut": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P], True, True),
    "cc_largest_k": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P], True, True),
}

GRAD_ANY, GRAD_LINEAR, GRAD_BRICKED = -1, 0, 1
ROUTE_NONE, ROUTE_SCATTER, ROUTE_TILE, ROUTE_PACKED_SCATTER, ROUTE_REGION, ROUTE_DETERMINISTIC = -1, 0, 1, 2, 3, 4
GRAPH_CAP_ONE = 1 << 28
DIR_XP, DIR_XM, DIR_YP, DIR_YM, DIR_ZP, DIR_ZM = range(6)

# the CPU twin's render_bwd does not take the forward outputs (colour, depth, acc): it recomputes
_CPU_RENDER_BWD = [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32]

# point query: the HIP flavour takes (reuse_packed_grid, workspace, bytes, stream) after the common arguments
_QUERY_COMMON_FWD = [_GD, _P, C.c_int64, _P]
_QUERY_COMMON_BWD = [_GD, _P, C.c_int64, _P, _P, _P, C.c_int32]

HIP_ONLY = {
    "query_fwd": (C.c_int, _QUERY_COMMON_FWD + [C.c_int32, _P, C.c_size_t, _P]),
    "query_bwd": (C.c_int, _QUERY_COMMON_BWD + [C.c_int32, _P, C.c_size_t, _P]),
    "abi_version": (C.c_int, []),
    "strerror": (C.c_char_p, [C.c_int]),
    "device_check": (C.c_int, [C.c_char_p, C.c_size_t]),
    "workspace_bytes": (C.c_size_t, [_GD, _RC, C.c_int64]),
    # fused optimiser step of a grid (gradient stays in the workspace between the two calls)
    "render_bwd_acc": (C.c_int, [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(C.c_int32), _P, C.c_size_t, _P]),
    "render_bwd_acc_into": (C.c_int, [_GD, _RC, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_int32), _P, C.c_size_t, _P, C.c_size_t, _P]),
    "render_bwd_layout": (C.c_int, [_GD, _RC, C.c_int64]),
    "workspace_grad_offset": (C.c_size_t, [_GD]),
    "workspace_grad_bytes": (C.c_size_t, [_GD]),
    "grid_adam_step": (C.c_int, [_GD, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int64, C.c_int64, C.POINTER(VoxeGridRegularisers), _P, C.c_size_t, _P]),
    "render_route": (C.c_int, [_GD, _RC, C.c_int64]),
    "disparity_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "clock_probe": (C.c_int, [C.c_int32, C.POINTER(C.c_double), _P]),
    "region_debug_layout": (C.c_int, [_GD, _RC, C.c_int64, C.POINTER(C.c_int64)]),
    "recon_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "recon_step": (C.c_int, [_GD, _RC, C.POINTER(VoxeReconStep), _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "recon_prefetch_stats": (C.c_int, [C.POINTER(C.c_int64)]),
    "recon_prefetch": (C.c_int, [_GD, _RC, C.POINTER(VoxeReconStep), _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "attn_masked_l1_scratch_bytes": (C.c_size_t, []),
    "attn_masked_l1": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "attn_refine_scratch_bytes": (C.c_size_t, [_GD, C.c_int64]),
    "attn_refine_step": (C.c_int, [_GD, _RC, C.POINTER(VoxeAttnRefineStep), _P, _P, C.c_int64, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "dcl_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "tv_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "graphcut_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "cc_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "profile_enable": (C.c_int, [C.c_int32]),
    "profile_read": (C.c_int, [C.POINTER(VoxeProfile)]),
}

CPU_ONLY = {
    "query_fwd": (C.c_int, _QUERY_COMMON_FWD),
    "query_bwd": (C.c_int, _QUERY_COMMON_BWD),
    "num_threads": (C.c_int, []),
    "jitter_uniform": (C.c_float, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int32]),
}


def hip_symbols():
    """Every symbol include/voxe.h declares for libvoxe_hip.so."""
    return ["voxe_" + n for n in list(_COMMON) + list(HIP_ONLY)]


def cpu_symbols():
    """Every symbol include/voxe.h declares for the oracle."""
    return ["voxe_cpu_" + n for n in list(_COMMON) + list(CPU_ONLY)]


def declare(lib, prefix):
    """Install prototypes on `lib` for the given family ("voxe_" or "voxe_cpu_")."""
    cpu = prefix == "voxe_cpu_"
    for name, (res, args, ws, st) in _COMMON.items():
        fn = getattr(lib, prefix + name)
        a = list(args)
        if cpu and name == "render_bwd":
            a = list(_CPU_RENDER_BWD)
        if not cpu:
            if ws:
                a += [_P, C.c_size_t]
            if st:
                a += [_P]
        fn.restype = res
        fn.argtypes = a
    extra = CPU_ONLY if cpu else HIP_ONLY
    for name, (res, args) in extra.items():
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = list(args)
    return lib
