"""fem_shell_amd -- MI355X-native element-matrix + CSR-assembly path behind fem-shell's API.

Host side (this package) mirrors the reference's element classes, ElementFactory and
MeshAssembler for the hot path only; all arithmetic runs in hand-written HIP kernels
(csrc/femshell_hip.hip) reached through the C ABI declared in include/femshell_hip.h.
PyTorch is used only for device buffers and streams.
"""
from .material import IsotropicMaterial, Material, OrthotropicMaterial  # noqa: F401
from .laminate import (  # noqa: F401
    Laminate, Ply, StrengthProperties, compute_Q, compute_Qbar, compute_shear_Cbar,
    create_laminate_from_angles, create_symmetric_laminate,
)
from .properties import CompositeShellProperty, ShellProperty, build_prop_table  # noqa: F401

__all__ = [
    "IsotropicMaterial", "OrthotropicMaterial", "Material", "Ply", "Laminate",
    "StrengthProperties", "compute_Q", "compute_Qbar", "compute_shear_Cbar",
    "create_laminate_from_angles", "create_symmetric_laminate", "ShellProperty",
    "CompositeShellProperty", "build_prop_table",
]
