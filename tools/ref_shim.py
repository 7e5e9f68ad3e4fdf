"""Import shim for the reference's hot-path modules (oracle tooling; runs ONLY in the build
container where /root/reference exists -- never on the GPU box, never from the product path).

Bypasses the two package __init__ files that pull pyvista/gmsh/petsc4py so that
fem_shell.elements / core.laminate / core.material / core.properties import with NumPy alone.
"""
import os
import sys
import types

REF = "/root/reference/src/fem_shell"


def install():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: golden fixtures can only be generated "
                           "in the build container")
    if "fem_shell" in sys.modules and getattr(sys.modules["fem_shell"], "_shimmed", False):
        return

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m._shimmed = True
        sys.modules[name] = m

    ns("fem_shell", REF)
    ns("fem_shell.core", REF + "/core")
    ns("fem_shell.constitutive", REF + "/constitutive")
    mesh = types.ModuleType("fem_shell.core.mesh")  # real one needs pyvista/gmsh/meshio
    mesh.MeshElement = type("MeshElement", (), {})
    mesh.MeshModel = type("MeshModel", (), {})
    sys.modules["fem_shell.core.mesh"] = mesh
