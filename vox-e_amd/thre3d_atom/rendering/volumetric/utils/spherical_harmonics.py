"""Real spherical-harmonics constants (degree 0..3) used by the SH voxel grid.

Values are those of the reference's thre3d_atom/rendering/volumetric/utils/spherical_harmonics.py:33-52
(originally from PlenOctrees).  Evaluation happens inside the HIP kernels (voxe_device.hpp sh_basis);
`sh_basis_torch` is a host helper for tools that need the basis values (e.g. baking), not on the
render path.
"""
import torch
from torch import Tensor

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [
    -0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
    -0.4570457994644658, 1.445305721320277, -0.5900435899266435,
]


def num_sh_coefficients(degree: int) -> int:
    return (degree + 1) ** 2


def sh_basis_torch(degree: int, viewdirs: Tensor) -> Tensor:
    """[..., (degree+1)^2] basis values for unit directions [..., 3]."""
    if not 0 <= degree <= 3:
        raise ValueError("only SH degrees 0..3 are supported")
    x, y, z = viewdirs[..., 0], viewdirs[..., 1], viewdirs[..., 2]
    out = [torch.full_like(x, C0)]
    if degree > 0:
        out += [-C1 * y, C1 * z, -C1 * x]
    if degree > 1:
        xx, yy, zz = x * x, y * y, z * z
        out += [C2[0] * x * y, C2[1] * y * z, C2[2] * (2.0 * zz - xx - yy), C2[3] * x * z, C2[4] * (xx - yy)]
    if degree > 2:
        out += [
            C3[0] * y * (3 * xx - yy), C3[1] * x * y * z, C3[2] * y * (4 * zz - xx - yy),
            C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy),
            C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy),
        ]
    return torch.stack(out, dim=-1)


def evaluate_spherical_harmonics(degree: int, sh_coeffs: Tensor, viewdirs: Tensor) -> Tensor:
    """Host helper with the reference's name and argument meaning (spherical_harmonics.py:64-132):
    sh_coeffs [..., C, (degree+1)^2], unit viewdirs [..., 3] -> [..., C].  The renderer never calls it (the kernels
    evaluate the basis per ray, voxe_device.hpp sh_basis); it exists for tools that bake or inspect SH grids."""
    if sh_coeffs.shape[-1] < num_sh_coefficients(degree):
        raise ValueError(f"degree {degree} needs {num_sh_coefficients(degree)} coefficients, got {sh_coeffs.shape[-1]}")
    basis = sh_basis_torch(degree, viewdirs)                      # [..., n]
    return (sh_coeffs[..., : basis.shape[-1]] * basis.unsqueeze(-2)).sum(dim=-1)
