"""ctypes binding of libibhip.so (the C ABI declared in include/ibhip.h).

There is no CPU fallback: if the shared library is missing, or a call returns
an error code, an exception is raised.
"""
import ctypes as C
import os

_here = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IBHIP_LIB") or os.path.join(_here, "libibhip.so")  # IBHIP_LIB: A/B builds

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
c_vp = C.c_void_p
c_i64 = C.c_int64
c_int = C.c_int


class IbhError(RuntimeError):
    pass


class ibh_fluid(C.Structure):
    _fields_ = [("R", C.c_float), ("gamma", C.c_float), ("mu_ref", C.c_float), ("Tref", C.c_float),
                ("S", C.c_float), ("nk", C.c_int32), ("k", C.c_float * 4)]


_SIGS = {
    "ibh_init": [c_int],
    "ibh_set_stream": [c_vp],
    "ibh_sync": [],
    "ibh_version": [],
    "ibh_malloc": [C.POINTER(c_vp), C.c_size_t],
    "ibh_free": [c_vp],
    "ibh_h2d": [c_vp, c_vp, C.c_size_t],
    "ibh_d2h": [c_vp, c_vp, C.c_size_t],
    "ibh_memset": [c_vp, c_int, C.c_size_t],
    "ibh_partition_create": [C.POINTER(c_vp), c_int, C.c_int32, c_vp, c_vp, c_vp,
                             C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_vp),
                             C.POINTER(c_vp), C.POINTER(c_vp), C.c_int32, c_vp, c_vp, c_int, c_int],
    "ibh_partition_destroy": [c_vp],
    "ibh_partition_info": [c_vp, C.POINTER(c_i64), c_int],
    "ibh_analyze2_host": [C.POINTER(c_vp), C.c_int32, c_vp, c_vp, C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_vp),
                          C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_vp), C.c_int32, c_vp, c_vp, c_int],
    "ibh_host2d_get": [c_vp, c_int, c_int, c_vp, c_i64, C.POINTER(c_i64)],
    "ibh_host2d_destroy": [c_vp],
    "ibh_at_owners": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_at_neighbors": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_at_faces": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_green_gauss": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64, c_int],
    "ibh_cell_gradient": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_cell_gradient_all": [c_vp, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_cell_gradient_nd": [c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_i64],
    "ibh_cell_gradient_fields": [c_vp, c_vp, c_int, c_i64, c_vp],
    "ibh_face_distance": [c_vp, c_int, c_vp],
    "ibh_owner_distance": [c_vp, c_int, c_vp],
    "ibh_neighbor_distance": [c_vp, c_int, c_vp],
    "ibh_face_gradient": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_jst_sensor": [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_muscl": [c_vp, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_i64],
    "ibh_acc_create": [C.POINTER(c_vp), C.c_int32, C.c_int32, c_vp, c_vp, c_vp, c_int],
    "ibh_acc_destroy": [c_vp],
    "ibh_accumulate": [c_vp, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_accumulate_diff_add": [c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_bc_create": [C.POINTER(c_vp), C.c_int32, c_vp, c_vp, c_vp, C.c_int32, c_vp, c_vp, c_vp, c_vp, c_int],
    "ibh_bc_destroy": [c_vp],
    "ibh_bc_interp": [c_vp, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_bc_blend": [c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "ibh_bc_apply": [c_vp, c_vp, c_int, c_i64, c_int, c_vp],
    "ibh_gather_rows": [c_vp, C.c_int32, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_scatter_rows": [c_vp, C.c_int32, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_copy_rows": [c_vp, c_vp, C.c_int32, c_vp, c_int, c_i64, c_vp, c_i64],
    "ibh_residual_advection": [c_vp, c_vp, c_vp, c_i64, c_vp, c_int],
    "ibh_residual_advection_n": [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int],
    "ibh_shear_rate_of_velocity": [c_vp, c_vp, c_i64, c_vp],
    "ibh_shear_rate_of_velocity_grad": [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64],
    "ibh_wray_agarwal_of": [c_vp, c_vp, c_vp, C.c_float, C.c_float, C.c_float, c_vp, c_vp, c_vp],
    "ibh_scalar_transport": [c_vp, c_vp, c_vp, C.c_float, c_vp, c_i64, c_vp, c_vp],
    "ibh_bcset_create": [C.POINTER(c_vp), c_int, C.POINTER(c_vp), c_vp, c_vp],
    "ibh_bcset_destroy": [c_vp],
    "ibh_bcset_info": [c_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "ibh_bcset_apply": [c_vp, c_vp],
    "ibh_timestep_advection": [c_vp, c_vp, c_i64, C.c_float, c_vp],
    "ibh_update_dev": [c_i64, c_vp, c_vp, c_vp, c_vp],
    "ibh_step_advection": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp],
    "ibh_step_advection_dt": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, C.c_float, c_vp],
    "ibh_ew_binary": [c_int, c_i64, c_int, c_vp, c_int, C.c_float, c_vp, c_int, C.c_float, c_vp],
    "ibh_ew_unary": [c_int, c_i64, c_vp, c_vp],
    "ibh_ew_fill": [c_i64, C.c_float, c_vp],
    "ibh_ew_reduce": [c_int, c_i64, c_vp, c_vp],
    "ibh_ew_eval": [c_i64, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp],
    "ibh_set_tuning": [C.c_char_p, c_int],
    "ibh_debug_buffer": [c_vp],
    "ibh_probe_dispatch": [c_int, c_int, c_int],
    "ibh_probe_sweep": [c_vp, c_vp, c_vp, c_i64, c_vp, c_int],
    "ibh_residual_euler_hll": [c_vp, c_vp, c_i64, c_vp, c_i64, C.POINTER(ibh_fluid), c_int],
    "ibh_cfd_speed_of_sound": [C.POINTER(ibh_fluid), c_i64, c_vp, c_vp],
    "ibh_cfd_dynamic_viscosity": [C.POINTER(ibh_fluid), c_i64, c_vp, c_vp],
    "ibh_cfd_heat_conductivity": [C.POINTER(ibh_fluid), c_i64, c_vp, c_vp],
    "ibh_cfd_primitive2state": [C.POINTER(ibh_fluid), c_int, c_i64, c_vp, c_i64, c_vp, c_i64],
    "ibh_cfd_state2primitive": [C.POINTER(ibh_fluid), c_int, c_i64, c_vp, c_i64, c_vp, c_i64],
    "ibh_cfd_inviscid_fluxes_hll": [C.POINTER(ibh_fluid), c_int, c_int, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64],
    "ibh_cfd_inviscid_fluxes_sensor": [C.POINTER(ibh_fluid), c_int, c_int, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp,
                                       c_vp, c_i64],
    "ibh_cfd_jst_sensor3": [c_i64, c_vp, c_vp, c_vp, c_vp],
    "ibh_cfd_shock_sensor": [c_int, c_i64, C.POINTER(c_vp), c_vp],
    "ibh_cfd_viscous_fluxes": [C.POINTER(ibh_fluid), c_int, c_int, c_i64, c_vp, c_i64, C.POINTER(c_vp), c_i64, c_vp,
                               C.c_float, c_vp, c_i64],
    "ibh_viscous_residual": [c_vp, C.POINTER(ibh_fluid), c_vp, c_i64, C.POINTER(c_vp), c_i64, c_int, c_vp, c_vp, c_i64],
    "ibh_cfd_flow_bc": [C.POINTER(ibh_fluid), c_int, c_i64, c_vp, c_i64, c_vp, c_i64, C.c_float, C.c_float, c_vp, c_int,
                        c_vp, c_vp, C.c_float, c_vp, c_vp, c_i64],
    "ibh_ipc_alloc": [C.POINTER(c_vp), C.c_size_t, c_int],
    "ibh_ipc_free": [c_vp],
    "ibh_ipc_export": [c_vp, c_vp],
    "ibh_ipc_import": [c_vp, C.POINTER(c_vp)],
    "ibh_ipc_close": [c_vp],
    "ibh_flag_signal": [c_vp, c_vp, c_int],
    "ibh_flag_wait": [c_vp, c_vp, c_int, C.c_uint32, c_vp],
    "ibh_halo_push": [c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_vp, c_vp],
    "ibh_halo_pull": [c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, C.c_uint32],
    "ibh_halo_exchange": [c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp,
                          c_vp, C.c_uint32],
    "ibh_step_advection_xgmi": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                c_vp, c_vp, c_vp, C.c_uint32, c_vp],
    "ibh_axpy_clamped": [c_i64, C.c_float, c_vp, c_vp],
    "ibh_axpy_clamped_sumsq": [c_i64, C.c_float, c_vp, c_vp, c_vp],
    "ibh_fas_update": [c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp],
    "ibh_axpy": [c_i64, C.c_float, c_vp, c_vp],
    "ibh_sumsq": [c_i64, c_vp, c_vp],
    "ibh_turb_wall_function_rey": [c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ibh_turb_wall_function": [c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ibh_turb_shear_rate": [c_int, c_i64, c_vp, c_vp],
    "ibh_turb_smagorinsky": [c_i64, c_vp, c_vp, C.c_float, c_vp],
    "ibh_turb_k_epsilon": [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "ibh_turb_wray_agarwal": [c_int, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, C.c_float, C.c_float, C.c_float, c_vp,
                              c_vp, c_vp],
    "ibh_turb_ducros": [c_int, c_i64, c_vp, c_vp],
    "ibh_turb_wale": [c_i64, c_vp, c_vp, C.c_float, c_vp],
    "ibh_pi_rademacher": [c_i64, C.c_uint64, c_vp],
    "ibh_pi_perturb": [c_i64, c_vp, c_vp, C.c_float, c_vp],
    "ibh_pi_fd": [c_i64, c_vp, c_vp, C.c_float, c_vp],
    "ibh_pi_hutch_accum": [c_i64, c_int, c_vp, c_vp, c_vp, C.c_float, c_vp],
    "ibh_pi_div_scalar": [c_i64, C.c_float, c_vp],
    "ibh_pi_invert_blocks": [c_i64, c_int, c_vp],
    "ibh_pi_apply_blocks": [c_i64, c_int, c_vp, c_vp, c_vp],
    "ibh_dot": [c_i64, c_vp, c_vp, c_vp],
    "ibh_maxabs": [c_i64, c_vp, c_vp],
    "ibh_pi_update": [c_i64, c_vp, C.c_float, c_vp, c_vp, c_vp, c_vp],
    "ibh_pi_normalize": [c_i64, c_vp, c_vp, C.c_float, c_vp],
}

EXPORTS = sorted(list(_SIGS) + ["ibh_last_error"])

_lib = None


def load():
    """Load libibhip.so (no GPU needed for loading); raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IbhError(
            f"{LIB_PATH} not found: the HIP library is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C immersedboundary.jl_amd/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("IBHIP_LIB"):   # A/B against an older build: entries it lacks fail when they are called
                continue
            raise IbhError(f"{LIB_PATH} does not export {name}: stale build")
        fn.argtypes = args
        fn.restype = c_int
    lib.ibh_last_error.argtypes = []
    lib.ibh_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def call(name, *args):
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise IbhError(f"{name} failed (code {rc}): {lib.ibh_last_error().decode()}")
    return rc
