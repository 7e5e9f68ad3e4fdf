"""Element classes with the reference's constructor signatures and members,
delegating every matrix to the batched HIP kernels (batch of 1).  Boundary types only: the bulk
path (MeshAssembler) never instantiates these.

Mirrors elements/elements.py:11-259 (ElementFamily, FemElement, ShellElement, ElementFactory),
elements/MITC4.py:26-99 and elements/MITC4_composite.py:34-120 of the reference.
"""
from enum import IntEnum
from functools import cached_property

import numpy as np

from .laminate import Laminate
from .material import OrthotropicMaterial
from .properties import CompositeShellProperty, build_prop_table, iso_row, laminate_row


class ElementFamily(IntEnum):
    SHELL = 2
    PLANE = 3
    SOLID = 4


class FemElement:
    _id_counter: int = 0
    vector_form = {}

    def __init__(self, name, node_coords, node_ids, material, dofs_per_node):
        self.name = name
        self.node_coords = node_coords
        self.node_ids = node_ids
        self.material = material
        self.dofs_per_node = dofs_per_node
        self.node_count = len(self.node_coords)
        self.dofs_count = self.node_count * self.dofs_per_node
        self.element_family = None
        self.id = FemElement._id_counter
        self.integration_order = 1
        FemElement._id_counter += 1

    @property
    def spatial_dimmension(self):  # (sic) elements/elements.py:41-49
        return 3 if self.element_family in (ElementFamily.SHELL, ElementFamily.SOLID) else 2

    def __repr__(self):
        return f"<Element id={self.id} name={self.name}>"


def _check_status(st, what):
    if st:
        raise ValueError(f"{what}: degenerate element geometry (status bits {int(st)})")


class ShellElement(FemElement):
    """Shared plumbing of the 6-DOF/node shells: one-element batches through fs_shell_quad_batch /
    fs_shell_tri_batch / fs_shell_kg_batch."""
    _NN = 0

    def __init__(self, name, node_coords, node_ids, material, dofs_per_node, thickness, nonlinear):
        super().__init__(name, node_coords, node_ids, material, dofs_per_node)
        self.thickness = thickness
        self.nonlinear = nonlinear
        self.element_family = ElementFamily.SHELL

    def _device_inputs(self):
        import torch
        from . import batch
        row = self._prop_row()
        props = np.zeros(1, dtype=row.dtype)
        props[0] = row
        dev = batch._dev()
        xyz = torch.from_numpy(self._initial_coords[None]).to(dev)
        pid = torch.zeros(1, dtype=torch.int32, device=dev)
        return xyz, pid, batch.to_device_props(props), dev

    def _run(self, **want):
        from . import batch
        xyz, pid, props, _ = self._device_inputs()
        out = batch.shell_batch(self._NN, xyz, pid, props, 1, dense=True, **want)
        _check_status(int(out["status"][0]), self.name)
        return out

    def update_configuration(self, displacements):
        """elements/MITC4.py:1067-1084, MITC3.py:1144-1187 (the directors they also update are
        unused by K_T / f_int)."""
        n = 6 * self._NN
        if len(displacements) != n:
            raise ValueError(f"Expected {n} DOFs, got {len(displacements)}")
        self._current_displacements = np.array(displacements, dtype=float)
        self._current_coords = self._initial_coords + self._current_displacements.reshape(self._NN, 6)[:, :3]

    def reset_configuration(self):
        self._current_displacements = np.zeros(6 * self._NN)
        self._current_coords = self._initial_coords.copy()

    # constitutive matrices as the kernels receive them (elements/MITC4.py:671-695, MITC4_composite.py:172-182)
    def Cm(self) -> np.ndarray:
        return np.array(self._prop_row()["A"], dtype=float).reshape(3, 3)

    def Cb(self) -> np.ndarray:
        return np.array(self._prop_row()["D"], dtype=float).reshape(3, 3)

    def Cs(self) -> np.ndarray:
        return np.array(self._prop_row()["Cs"], dtype=float).reshape(2, 2)

    def _run_nl(self, want_KT, want_F):
        import torch
        from . import batch
        u = torch.from_numpy(self._current_displacements[None].copy()).to(batch._dev())
        return self._run(want_K=False, u_e=u, want_KT=want_KT, want_F=want_F)

    def _frame_coords(self):
        """Coordinates the reference's T() builds its frame from: the initial ones for MITC4 (MITC4.py:634-640 reads the
        frame of __init__), the CURRENT ones for MITC3 (MITC3.py:471-492)."""
        import torch
        from . import batch
        c = self._initial_coords if self._NN == 4 else self._current_coords
        return torch.from_numpy(np.ascontiguousarray(c[None])).to(batch._dev())

    def _to_local(self, K=None, f=None):
        from . import batch
        Kl, fl = batch.shell_rotate_local(self._NN, self._frame_coords(), K, f)
        return (None if Kl is None else np.ascontiguousarray(Kl[0].cpu().numpy()),
                None if fl is None else np.ascontiguousarray(fl[0].cpu().numpy()))

    def compute_tangent_stiffness(self, sigma=None, transform_to_global=True):
        """K_T = T^T sym(K0 + K_L + K_sigma) T (elements/MITC4.py:1246-1298, MITC3.py:1717-1777, composite overrides).
        sigma: local membrane stress [sxx, syy, sxy] used for K_sigma instead of the one derived from the current
        displacements; transform_to_global=False returns the local-axes matrix T K_T T^T."""
        import torch
        from . import batch
        if sigma is None:
            KT = self._run_nl(True, False)["KT"]
        else:
            xyz, pid, props, dev = self._device_inputs()
            sg = torch.from_numpy(np.ascontiguousarray(np.asarray(sigma, dtype=np.float64).reshape(1, 3))).to(dev)
            u = torch.from_numpy(self._current_displacements[None].copy()).to(dev)
            KT, st = batch.shell_kt_sigma_batch(self._NN, xyz, pid, props, 1, u, sg)
            _check_status(int(st[0]), self.name)
        if not transform_to_global:
            return self._to_local(K=KT)[0]
        return np.ascontiguousarray(KT[0].cpu().numpy())

    def compute_internal_forces(self, transform_to_global=True):
        """elements/MITC4.py:1300-1341, MITC3.py:1779-1832; transform_to_global=False returns T f."""
        f = self._run_nl(False, True)["f_int"]
        if not transform_to_global:
            return self._to_local(f=f)[1]
        return np.ascontiguousarray(f[0].cpu().numpy())

    def compute_residual(self, f_ext, transform_to_global=True):
        return f_ext - self.compute_internal_forces(transform_to_global=transform_to_global)

    def compute_geometric_stiffness(self, sigma_membrane, transform_to_global=True):
        """K_sigma from a LOCAL membrane stress [sxx, syy, sxy] (elements/MITC4.py:1191-1225,
        MITC3.py:1501-1580); transform_to_global=False returns the local-axes matrix."""
        import torch
        from . import batch
        xyz, pid, props, dev = self._device_inputs()
        sg = np.asarray(sigma_membrane, dtype=np.float64)
        if sg.shape == (2, 2):
            sg = np.array([sg[0, 0], sg[1, 1], sg[0, 1]])
        sg = torch.from_numpy(np.ascontiguousarray(sg.reshape(1, 3))).to(dev)
        KG = batch.shell_kg_batch(self._NN, xyz, pid, props, 1, sg, dense=True)
        if not transform_to_global:
            return self._to_local(K=KG)[0]
        return np.ascontiguousarray(KG[0].cpu().numpy())


class MITC4(ShellElement):
    """MITC4+ quadrilateral shell, 6 DOF/node (elements/MITC4.py:26-99)."""

    vector_form = {"U": ("Ux", "Uy", "Uz"), "θ": ("θx", "θy", "θz")}
    _NN = 4

    def __init__(self, node_coords, node_ids, material, thickness, shear_correction_factor=None,
                 nonlinear=False):
        super().__init__("MITC4", node_coords=node_coords, node_ids=node_ids, material=material,
                         dofs_per_node=6, thickness=thickness, nonlinear=nonlinear)
        self.element_type = "MITC4"
        self._initial_coords = np.array(node_coords, dtype=float).copy()
        if self._initial_coords.shape != (4, 3):
            raise ValueError("MITC4 needs 4 nodes with 3 coordinates each")
        self._current_displacements = np.zeros(24)
        self._current_coords = self._initial_coords.copy()
        self._kappa_arg = shear_correction_factor

    def _prop_row(self):
        return iso_row(self.material, self.thickness, self._kappa_arg, self.nonlinear)

    @cached_property
    def K(self) -> np.ndarray:
        """Global-axes stiffness [24,24] (elements/MITC4.py:1018-1036)."""
        return np.ascontiguousarray(self._run(want_K=True)["K"][0].cpu().numpy())

    @cached_property
    def M(self) -> np.ndarray:
        """Consistent mass [24,24] (elements/MITC4.py:1038-1061)."""
        return np.ascontiguousarray(self._run(want_K=False, want_M=True)["M"][0].cpu().numpy())


class MITC4Composite(MITC4):
    """Laminated MITC4+ (elements/MITC4_composite.py:34-120)."""

    def __init__(self, node_coords, node_ids, laminate: Laminate, kx_mod=1.0, ky_mod=1.0,
                 nonlinear=False):
        self.laminate = laminate
        self._kx_mod, self._ky_mod = kx_mod, ky_mod
        eq = laminate.get_equivalent_properties()
        ref = laminate.plies[0].material
        Ex, Ey, Gxy, nuxy = eq["Ex_membrane"], eq["Ey_membrane"], eq["Gxy_membrane"], eq["nuxy_membrane"]
        equivalent = OrthotropicMaterial(
            name=f"Equivalent_{laminate}", E=(Ex, Ey, Ey), G=(Gxy, ref.G[1], ref.G[2]),
            nu=(nuxy, nuxy * Ey / Ex, nuxy), rho=ref.rho,
            shear_correction_factor=laminate.shear_correction_factor)
        super().__init__(node_coords=node_coords, node_ids=node_ids, material=equivalent,
                         thickness=laminate.total_thickness,
                         shear_correction_factor=laminate.shear_correction_factor,
                         nonlinear=nonlinear)
        self.element_type = "MITC4Composite"

    A = property(lambda self: self.laminate.A)
    B_coup = property(lambda self: self.laminate.B)
    D = property(lambda self: self.laminate.D)
    has_coupling = property(lambda self: not self.laminate.is_symmetric)

    def _prop_row(self):
        return laminate_row(self.laminate, self.nonlinear)


class MITC3(ShellElement):
    """MITC3+ triangular shell, 6 DOF/node (elements/MITC3.py:35-210)."""

    vector_form = {"U": ("Ux", "Uy", "Uz"), "θ": ("θx", "θy", "θz")}
    _NN = 3

    def __init__(self, node_coords, node_ids, material, thickness, shear_correction_factor=None,
                 nonlinear=False):
        super().__init__("MITC3", node_coords=node_coords, node_ids=node_ids, material=material,
                         dofs_per_node=6, thickness=thickness, nonlinear=nonlinear)
        self.element_type = "MITC3"
        self._initial_coords = np.array(node_coords, dtype=float).copy()
        if self._initial_coords.shape != (3, 3):
            raise ValueError("MITC3 needs 3 nodes with 3 coordinates each")
        self._kappa_arg = shear_correction_factor
        self._current_displacements = np.zeros(18)
        self._current_coords = self._initial_coords.copy()
        # MITC3.py:375-418 raises at construction for a degenerate triangle
        v12, v13 = self._initial_coords[1] - self._initial_coords[0], self._initial_coords[2] - self._initial_coords[0]
        if np.linalg.norm(np.cross(v12, v13)) < 1e-12:
            raise ValueError("Degenerate triangle: nodes are collinear")

    def _prop_row(self):
        if isinstance(self.material, OrthotropicMaterial):
            raise NotImplementedError("Orthotropic material not yet supported in MITC3")
        return iso_row(self.material, self.thickness, self._kappa_arg, self.nonlinear)

    @cached_property
    def K(self) -> np.ndarray:
        """Global-axes stiffness [18,18] (elements/MITC3.py:1077-1103)."""
        return np.ascontiguousarray(self._run(want_K=True)["K"][0].cpu().numpy())

    @cached_property
    def M(self) -> np.ndarray:
        """Consistent mass [18,18] (elements/MITC3.py:1105-1138)."""
        return np.ascontiguousarray(self._run(want_K=False, want_M=True)["M"][0].cpu().numpy())

    def area(self) -> float:
        """elements/MITC3.py:427-447."""
        x = self._initial_coords
        return 0.5 * float(np.linalg.norm(np.cross(x[1] - x[0], x[2] - x[0])))

    def compute_centrifugal_prestress(self, omega, rotation_axis, rotation_center) -> np.ndarray:
        """elements/MITC3.py:1633-1715: local membrane stress [sxx, syy, sxy] of a rotating structure."""
        from . import batch
        xyz, pid, props, _ = self._device_inputs()
        return np.ascontiguousarray(batch.centrifugal_prestress(False, 3, xyz, pid, props, 1, omega,
                                                                rotation_axis, rotation_center)[0].cpu().numpy())


class MITC3Composite(MITC3):
    """Laminated MITC3+ (elements/MITC3_composite.py:34-135)."""

    def __init__(self, node_coords, node_ids, laminate: Laminate, nonlinear=False):
        self.laminate = laminate
        eq = laminate.get_equivalent_properties()
        ref = laminate.plies[0].material
        Ex, Ey, Gxy, nuxy = eq["Ex_membrane"], eq["Ey_membrane"], eq["Gxy_membrane"], eq["nuxy_membrane"]
        equivalent = OrthotropicMaterial(
            name=f"Equivalent_{laminate}", E=(Ex, Ey, Ey), G=(Gxy, ref.G[1], ref.G[2]),
            nu=(nuxy, nuxy * Ey / Ex, nuxy), rho=ref.rho,
            shear_correction_factor=laminate.shear_correction_factor)
        super().__init__(node_coords=node_coords, node_ids=node_ids, material=equivalent,
                         thickness=laminate.total_thickness,
                         shear_correction_factor=laminate.shear_correction_factor,
                         nonlinear=nonlinear)
        self.element_type = "MITC3Composite"

    A = property(lambda self: self.laminate.A)
    B_coup = property(lambda self: self.laminate.B)
    D = property(lambda self: self.laminate.D)
    has_coupling = property(lambda self: not self.laminate.is_symmetric)

    def _prop_row(self):
        return laminate_row(self.laminate, self.nonlinear)


class SolidElement(FemElement):
    """Linear and quadratic 3-D solids, 3 DOF/node (elements/SOLID.py:47-692): K, M, body_load and
    compute_geometric_stiffness run through fs_solid_batch (batch of 1)."""

    vector_form = {"U": ("Ux", "Uy", "Uz")}
    _NN = 0

    def __init__(self, name, node_coords, node_ids, material, orientation=None, reduced_integration=False):
        super().__init__(name=name, node_coords=np.asarray(node_coords), node_ids=list(node_ids),
                         material=material, dofs_per_node=3)
        if len(node_ids) != self._NN:
            raise ValueError(f"{name} requires {self._NN} nodes, got {len(node_ids)}")
        self.element_family = ElementFamily.SOLID
        self.orientation = orientation
        self.reduced_integration = reduced_integration
        if self.node_coords.shape[1] != 3:
            raise ValueError(f"Solid elements require 3D coordinates, got shape {self.node_coords.shape}")

    @property
    def C(self) -> np.ndarray:
        from .properties import solid_C
        return solid_C(self.material, self.orientation)

    def _run(self, **kw):
        import torch
        from . import batch
        from .properties import solid_row
        dev = batch._dev()
        xyz = torch.from_numpy(np.ascontiguousarray(self.node_coords, dtype=np.float64)[None]).to(dev)
        pid = torch.zeros(1, dtype=torch.int32, device=dev)
        props = batch.to_device_solid_props(solid_row(self.material, self.orientation)[None])
        if kw.get("sigma") is not None:
            kw["sigma"] = torch.as_tensor(np.asarray(kw["sigma"], dtype=np.float64).reshape(1, 6), device=dev)
        out = batch.solid_batch(self._NN, xyz, pid, props, 1, dense=True,
                                reduced=bool(self.reduced_integration) and self._NN == 8, **kw)
        if int(out["status"][0]):
            raise ValueError(f"Non-positive Jacobian determinant in {self.name} element")
        return out

    @cached_property
    def K(self) -> np.ndarray:
        """elements/SOLID.py:543-567."""
        return np.ascontiguousarray(self._run(want_K=True)["K"][0].cpu().numpy())

    @cached_property
    def M(self) -> np.ndarray:
        """elements/SOLID.py:569-597."""
        return np.ascontiguousarray(self._run(want_K=False, want_M=True)["M"][0].cpu().numpy())

    def body_load(self, body_force) -> np.ndarray:
        """elements/SOLID.py:599-631."""
        return np.ascontiguousarray(self._run(want_K=False, body=np.asarray(body_force)[:3])["f"][0].cpu().numpy())

    def compute_geometric_stiffness(self, sigma_membrane, transform_to_global=True) -> np.ndarray:
        """elements/SOLID.py:322-374 (Voigt stress xx,yy,zz,xy,yz,zx in global axes)."""
        return np.ascontiguousarray(self._run(want_K=False, sigma=sigma_membrane)["KG"][0].cpu().numpy())

    def compute_centrifugal_prestress(self, omega, rotation_axis, rotation_center) -> np.ndarray:
        """elements/SOLID.py:268-320: global Voigt stress [xx,yy,zz,xy,yz,zx]."""
        import torch
        from . import batch
        from .properties import solid_row
        dev = batch._dev()
        xyz = torch.from_numpy(np.ascontiguousarray(self.node_coords, dtype=np.float64)[None]).to(dev)
        pid = torch.zeros(1, dtype=torch.int32, device=dev)
        props = batch.to_device_solid_props(solid_row(self.material, self.orientation)[None])
        return np.ascontiguousarray(batch.centrifugal_prestress(True, self._NN, xyz, pid, props, 1, omega,
                                                                rotation_axis, rotation_center)[0].cpu().numpy())


def _solid_class(name, nn, doc):
    def __init__(self, node_coords, node_ids, material, orientation=None, reduced_integration=False):
        SolidElement.__init__(self, name, node_coords, node_ids, material, orientation,
                              reduced_integration if nn == 8 else False)
    return type(name, (SolidElement,), {"__init__": __init__, "_NN": nn, "__doc__": doc})


TETRA4 = _solid_class("TETRA4", 4, "4-node tetrahedron, 1-point rule (elements/SOLID.py:1545-1597).")
PYRAMID5 = _solid_class("PYRAMID5", 5, "5-node pyramid, Felippa 8-point rule (elements/SOLID.py:1067-1218).")
WEDGE6 = _solid_class("WEDGE6", 6, "6-node wedge, 3x2 rule with the reference's x2 weights (elements/SOLID.py:700-829).")
HEXA8 = _solid_class("HEXA8", 8, "8-node brick, 2x2x2 (or 1-point reduced) rule (elements/SOLID.py:1716-1818).")
TETRA10 = _solid_class("TETRA10", 10, "10-node tetrahedron (Gmsh order), 4-point rule (elements/SOLID.py:1600-1708).")
PYRAMID13 = _solid_class("PYRAMID13", 13, "13-node quadratic pyramid (Gmsh order), 18-point Felippa rule, rational serendipity "
                         "functions of the scaled base (elements/SOLID.py:1221-1537).")
WEDGE15 = _solid_class("WEDGE15", 15, "15-node wedge (Gmsh order), 7x3 rule with the x2 weights (elements/SOLID.py:832-1059).")
HEXA20 = _solid_class("HEXA20", 20, "20-node serendipity brick (Gmsh order), 3x3x3 rule; derivatives are the reference's "
                      "forward differences, h = 1e-8 (elements/SOLID.py:1821-1946).")


class ElementFactory:
    """elements/elements.py:107-259: dispatch by family x node count x composite."""

    @staticmethod
    def get_element(element_family, mesh_element, shell_property=None, **kwargs):
        if shell_property is not None:
            prop_kwargs = shell_property.to_element_kwargs()
            is_composite = isinstance(shell_property, CompositeShellProperty)
            laminate = prop_kwargs.pop("laminate", None)
            kwargs.update(prop_kwargs)
        else:
            laminate = kwargs.pop("laminate", None)
            is_composite = laminate is not None
        shell_map = {3: MITC3, 4: MITC4}
        shell_composite_map = {3: MITC3Composite, 4: MITC4Composite}
        node_ids = mesh_element.node_ids
        node_coords = mesh_element.node_coords
        node_count = mesh_element.node_count
        try:
            if element_family == ElementFamily.SHELL:
                if is_composite:
                    return shell_composite_map[node_count](node_coords=node_coords, node_ids=node_ids,
                                                           laminate=laminate, **kwargs)
                return shell_map[node_count](node_coords=node_coords, node_ids=node_ids, **kwargs)
            if element_family == ElementFamily.SOLID:
                solid_map = {4: TETRA4, 5: PYRAMID5, 6: WEDGE6, 8: HEXA8, 10: TETRA10, 13: PYRAMID13, 15: WEDGE15,
                             20: HEXA20}
                kwargs.pop("thickness", None)
                return solid_map[node_count](node_coords=node_coords, node_ids=node_ids, **kwargs)
            raise NotImplementedError(f"Element family {element_family} not implemented")
        except KeyError:
            return False
