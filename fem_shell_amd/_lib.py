"""ctypes binding of libfemshell_hip.so (include/femshell_hip.h). Fails loudly when the HIP
library is missing -- there is NO CPU fallback on the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEMSHELL_HIP_LIB") or os.path.join(_HERE, "csrc", "libfemshell_hip.so")

FS_WANT_K, FS_WANT_M, FS_WANT_KT, FS_WANT_FINT = 1, 2, 4, 8
FS_WANT_FBODY, FS_WANT_KG = 16, 32
FS_LAYOUT_DENSE, FS_LAYOUT_SYMBLK, FS_LAYOUT_SYMBLK_T16 = 0, 1, 2
FS_STRESS_MEMBRANE, FS_STRESS_BENDING = 1, 2
FS_SOLID_STRESS_GP, FS_SOLID_STRESS_MEAN, FS_SOLID_STRESS_NODES = 0, 1, 2
FS_INERTIAL_CENTRIFUGAL, FS_INERTIAL_CORIOLIS, FS_INERTIAL_EULER = 1, 2, 4
FS_FUSED_HAS_COUPLED = 1
FS_FUSED_MASS = 2
FS_TANGENT_TWO_WGS_PER_CU = 1
FS_FUSED_MAX_PATCH_ELEMS = 64
FS_HEX_MAX_PATCH_ELEMS, FS_HEX_MAX_PATCH_NODES, FS_HEX_MAX_PATCH_LANES = 100, 64, 512

_vp, _i64, _i32, _u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32

SIGNATURES = {
    "fs_last_error": (ctypes.c_char_p, []),
    "fs_device_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int)]),
    "fs_shell_quad_batch": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _u32, _u32, _vp, _vp,
                                           _vp, _vp, _vp, _vp]),
    "fs_shell_quad_fint_noalloc": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp]),
    "fs_fused_assemble_quad": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _u32,
                                              _vp, _vp, _vp]),
    "fs_fused_assemble_quad_tangent": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32,
                                                      _vp, _vp, _vp, _vp, _u32, _vp]),
    "fs_fused_tangent_offsets": (ctypes.c_int, [_i64, _vp, _vp, ctypes.POINTER(_i64), _vp]),
    "fs_fused_tangent_pack": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "fs_shell_quad_fint_nl": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "fs_fused_plan_check": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp]),
    "fs_shell_quad_geom_status": (ctypes.c_int, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "fs_fused_assemble_hex8": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                              _vp, _vp, _vp]),
    "fs_fused_hex_plan_check": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "fs_fused_hex_plan_build": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_hex_plan_export": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_hex_plan_free": (None, [_vp]),
    "fs_fused_assemble_hex8_mass": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_assemble_tri": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_assemble_tri_mass": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_tri_plan_check": (ctypes.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "fs_fused_tri_plan_build": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_tri_plan_export": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_tri_plan_free": (None, [_vp]),
    "fs_fused_plan_build": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "fs_fused_plan_export": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_plan_free": (None, [_vp]),
    "fs_fused_plan_device_begin": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_plan_device_finish": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_fused_plan_device_free": (None, [_vp]),
    "fs_shell_tri_batch": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _u32, _u32, _vp, _vp,
                                          _vp, _vp, _vp, _vp]),
    "fs_solid_batch": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _i32, _i64, _u32, _u32, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp]),
    "fs_solid2_table": (ctypes.c_int, [_i32, _vp, _vp, _vp, ctypes.POINTER(_i32)]),
    "fs_shell_kg_batch": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _i64, _u32, _vp, _vp]),
    "fs_shell_kt_sigma_batch": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "fs_shell_rotate_local": (ctypes.c_int, [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fs_centrifugal_prestress": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _i32, ctypes.c_double, _vp, _vp,
                                                _i64, _vp, _vp]),
    "fs_shell_quad_geom": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fs_blk_expand": (ctypes.c_int, [_vp, _u32, _i64, _i32, _i32, _vp, _vp]),
    "fs_csr_numeric": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fs_vec_gather_sum": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fs_csr_numeric_desc": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "fs_csr_pattern_blk": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "fs_csr_symbolic_begin": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _u32, _i64, _i64, _vp, _i64,
                                             ctypes.POINTER(_vp), ctypes.POINTER(_i64), ctypes.POINTER(_i64), _vp]),
    "fs_csr_symbolic_finish": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_csr_symbolic_free": (None, [_vp]),
    "fs_csr_zero_rows": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_double, _vp, _vp, _vp]),
    "fs_csr_reduce_symbolic": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, ctypes.POINTER(_i64),
                                              ctypes.POINTER(_i64), _vp]),
    "fs_csr_reduce_fill": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fs_expand_solution": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fs_csr_row_sums": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "fs_csr_blk_matvec": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "fs_halo_pack": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "fs_halo_add": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "fs_gather_dofs": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "fs_bench_fp64_fma": (ctypes.c_int, [_i32, _i32, _vp, _vp]),
    "fs_inertial_forces": (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_double, _vp, _vp, _u32, _vp, _vp]),
    "fs_shell_stress_batch": (ctypes.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i32, _i32, _u32, _i32,
                                             _vp, _vp, _vp]),
    "fs_solid_stress_batch": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i32, _vp, _vp,
                                             _vp, _vp]),
    "fs_nodal_average": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fs_stress_invariants": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
}

_lib = None


class FemShellHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FemShellHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch ships its own libamdhip64 (same soname as /opt/rocm's): load torch FIRST so that this
    # library binds to the HIP runtime that owns torch's device buffers and streams -- two HIP
    # runtimes in one process do not share devices ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().fs_last_error().decode(errors="replace")
        raise FemShellHipError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
