"""ctypes mirror of include/idocp_hip.h.

This is the binding a Python caller of the drop-in library uses; the tests and
bench.py go through it so that every GPU parity test exercises the C ABI.  It
holds no arithmetic: structures, prototypes and thin argument marshalling only.
"""
import ctypes as C
import os

import numpy as np

MAX_JOINTS, MAX_NV, MAX_NQ, MAX_CONTACTS = 16, 24, 25, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libidocp_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Model(C.Structure):
    _fields_ = [
        ("njoints", C.c_int), ("nq", C.c_int), ("nv", C.c_int), ("nu", C.c_int),
        ("has_floating_base", C.c_int),
        ("parent", C.c_int * MAX_JOINTS), ("jtype", C.c_int * MAX_JOINTS),
        ("idx_q", C.c_int * MAX_JOINTS), ("idx_v", C.c_int * MAX_JOINTS),
        ("axis", (C.c_double * 3) * MAX_JOINTS),
        ("plc_R", (C.c_double * 9) * MAX_JOINTS), ("plc_p", (C.c_double * 3) * MAX_JOINTS),
        ("mass", C.c_double * MAX_JOINTS), ("com", (C.c_double * 3) * MAX_JOINTS),
        ("inertia", (C.c_double * 9) * MAX_JOINTS),
        ("gravity", C.c_double * 3),
        ("q_min", C.c_double * MAX_NV), ("q_max", C.c_double * MAX_NV),
        ("v_max", C.c_double * MAX_NV), ("u_max", C.c_double * MAX_NV),
        ("ncontacts", C.c_int),
        ("contact_frame_id", C.c_int * MAX_CONTACTS), ("contact_joint", C.c_int * MAX_CONTACTS),
        ("contact_R", (C.c_double * 9) * MAX_CONTACTS), ("contact_p", (C.c_double * 3) * MAX_CONTACTS),
        ("total_mass", C.c_double),
    ]


class TaskComponent(C.Structure):
    _fields_ = [("dim", C.c_int), ("joint", C.c_int), ("frame_R", C.c_double * 9), ("frame_p", C.c_double * 3),
                ("weight", C.c_double * 6), ("weightf", C.c_double * 6), ("weighti", C.c_double * 6), ("ref", C.c_double * 12)]


class Cost(C.Structure):
    _fields_ = [
        ("q_ref", C.c_double * MAX_NQ), ("v_ref", C.c_double * MAX_NV), ("u_ref", C.c_double * MAX_NV),
        ("q_weight", C.c_double * MAX_NV), ("v_weight", C.c_double * MAX_NV),
        ("a_weight", C.c_double * MAX_NV), ("u_weight", C.c_double * MAX_NV),
        ("qf_weight", C.c_double * MAX_NV), ("vf_weight", C.c_double * MAX_NV),
        ("f_weight", (C.c_double * 3) * MAX_CONTACTS), ("f_ref", (C.c_double * 3) * MAX_CONTACTS),
        ("use_trotting_ref", C.c_int),
        ("t_start", C.c_double), ("t_period", C.c_double), ("step_length", C.c_double),
        ("front_swing_knee", C.c_double), ("hip_swing_knee", C.c_double),
        ("front_stance_knee", C.c_double), ("hip_stance_knee", C.c_double),
        ("qi_weight", C.c_double * MAX_NV), ("vi_weight", C.c_double * MAX_NV), ("dvi_weight", C.c_double * MAX_NV),
        ("fi_weight", (C.c_double * 3) * MAX_CONTACTS), ("fi_ref", (C.c_double * 3) * MAX_CONTACTS),
        ("use_time_varying_ref", C.c_int), ("tv_t_begin", C.c_double), ("tv_t_end", C.c_double),
        ("task_dim", C.c_int), ("task_joint", C.c_int), ("task_frame_R", C.c_double * 9), ("task_frame_p", C.c_double * 3),
        ("task_weight", C.c_double * 6), ("task_weightf", C.c_double * 6), ("task_ref", C.c_double * 12), ("task_time_varying", C.c_int),
        ("task_weighti", C.c_double * 6),
        ("task_extra_count", C.c_int), ("task_extra", TaskComponent * 3),
    ]

    def add_task(self, dim, joint, frame_R, frame_p, weight, weightf, ref, weighti=None):
        """one more TaskSpace3DCost / TaskSpace6DCost component (idocp_cost_t::task_extra; the first one lives in the task_* fields)"""
        assert self.task_dim != 0 and self.task_extra_count < 3
        t = self.task_extra[self.task_extra_count]
        t.dim, t.joint = int(dim), int(joint)
        for name, vals in (("frame_R", frame_R), ("frame_p", frame_p), ("weight", weight), ("weightf", weightf), ("weighti", weighti if weighti is not None else weight), ("ref", ref)):
            arr = getattr(t, name)
            vals = np.asarray(vals, dtype=np.float64).ravel()
            for i in range(len(arr)):
                arr[i] = float(vals[i]) if i < len(vals) else 0.0
        self.task_extra_count += 1
        return self

    def set(self, name, values):
        arr = getattr(self, name)
        values = np.atleast_1d(np.asarray(values, dtype=np.float64))
        for i, x in enumerate(values):
            arr[i] = float(x)
        return self


class Constraints(C.Structure):
    _fields_ = [
        ("joint_position_limits", C.c_int), ("joint_velocity_limits", C.c_int),
        ("joint_torque_limits", C.c_int),
        ("linearized_friction_cone", C.c_int), ("mu", C.c_double),
        ("barrier", C.c_double), ("fraction_to_boundary_rate", C.c_double),
        ("linearized_impulse_friction_cone", C.c_int),
        ("friction_cone", C.c_int), ("impulse_friction_cone", C.c_int),
        ("joint_acceleration_lower_limit", C.c_int), ("joint_acceleration_upper_limit", C.c_int),
        ("a_min", C.c_double * MAX_NV), ("a_max", C.c_double * MAX_NV),
        ("contact_distance", C.c_int),
    ]


class LibraryMissing(RuntimeError):
    pass


_lib = None


def _proto(lib):
    vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
    P = C.POINTER
    lib.idocp_model_from_urdf.argtypes = [cs, P(ci), ci, P(Model)]
    lib.idocp_model_from_urdf.restype = ci
    lib.idocp_model_frame_id.argtypes = [cs, cs]
    lib.idocp_model_frame_id.restype = ci
    lib.idocp_model_frame_placement.argtypes = [cs, ci, P(ci), P(cd), P(cd)]
    lib.idocp_model_frame_placement.restype = ci
    lib.idocp_abi_check.argtypes = [C.c_ulong, C.c_ulong, C.c_ulong]
    lib.idocp_abi_check.restype = ci
    if lib.idocp_abi_check(C.sizeof(Model), C.sizeof(Cost), C.sizeof(Constraints)) != 0:
        raise LibraryMissing("idocp_amd/capi.py structs do not match libidocp_hip.so (rebuild: python idocp_amd/build.py)")
    lib.idocp_model_contact_positions.argtypes = [P(Model), vp, vp]
    lib.idocp_model_frame_world_placement.argtypes = [P(Model), vp, ci, vp, vp, vp, vp]
    lib.idocp_model_frame_world_placement.restype = ci
    lib.idocp_model_contact_positions.restype = ci
    lib.idocp_model_integrate_configuration.argtypes = [P(Model), vp, vp, cd, vp]
    lib.idocp_model_integrate_configuration.restype = ci
    lib.idocp_model_subtract_configuration.argtypes = [P(Model), vp, vp, vp]
    lib.idocp_model_subtract_configuration.restype = ci
    lib.idocp_model_normalize_configuration.argtypes = [P(Model), vp]
    lib.idocp_model_normalize_configuration.restype = ci
    lib.idocp_cost_init.argtypes = [P(Cost)]
    lib.idocp_cost_init.restype = None
    lib.idocp_constraints_init.argtypes = [P(Constraints)]
    lib.idocp_constraints_init.restype = None
    lib.idocp_last_error.restype = cs
    lib.idocp_version.restype = cs
    lib.idocp_unocp_create.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, P(vp)]
    lib.idocp_unocp_create.restype = ci
    lib.idocp_unparnmpc_create.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, P(vp)]
    lib.idocp_unparnmpc_create.restype = ci
    lib.idocp_unparnmpc_create_shard.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, ci, ci, P(vp)]
    lib.idocp_unparnmpc_create_shard.restype = ci
    lib.idocp_unocp_destroy.argtypes = [vp]
    lib.idocp_unocp_destroy.restype = None
    for name, args in [
        ("idocp_unocp_set_solution", [vp, cs, c_double_p]),
        ("idocp_unocp_set_solution_batch", [vp, cs, c_double_p]),
        ("idocp_unocp_set_solution_only", [vp, cs, c_double_p]),
        ("idocp_unocp_set_cost", [vp, P(Cost)]),
        ("idocp_unocp_init_constraints", [vp]),
        ("idocp_unocp_set_task_refs", [vp, c_double_p]),
        ("idocp_unocp_update_solution", [vp, cd, c_double_p, c_double_p, ci]),
        ("idocp_unocp_update_solution_device", [vp, cd, vp, vp]),
        ("idocp_unocp_synchronize", [vp]),
        ("idocp_device_alloc", [P(vp), C.c_ulonglong]),
        ("idocp_device_free", [vp]),
        ("idocp_device_upload", [vp, vp, C.c_ulonglong]),
        ("idocp_device_count", [P(ci)]),
        ("idocp_unocp_compute_kkt_residual", [vp, cd, c_double_p, c_double_p]),
        ("idocp_unocp_kkt_error", [vp, c_double_p]),
        ("idocp_unocp_get_solution", [vp, cs, ci, c_double_p]),
        ("idocp_unocp_get_direction", [vp, cs, ci, c_double_p]),
        ("idocp_unocp_get_step_sizes", [vp, c_double_p, c_double_p]),
        ("idocp_unocp_get_riccati", [vp, ci, c_double_p, c_double_p, c_double_p, c_double_p]),
        ("idocp_unocp_get_torque_feedback_gain", [vp, ci, ci, c_double_p, c_double_p]),
        ("idocp_unocp_get_constraint_data", [vp, ci, c_double_p, c_double_p]),
        ("idocp_unocp_dimc", [vp]),
        ("idocp_unocp_is_current_solution_feasible", [vp, c_int_p, c_int_p]),
        ("idocp_unocp_launch_linearize", [vp, cd, vp, vp]),
        ("idocp_unocp_clear_line_search_filter", [vp]),
        ("idocp_unocp_line_search_eval", [vp, c_double_p, c_double_p, c_double_p]),
        ("idocp_unparnmpc_init_backward_correction", [vp, cd]),
        ("idocp_unparnmpc_update_solution", [vp, cd, c_double_p, c_double_p, ci]),
        ("idocp_unparnmpc_update_solution_device", [vp, cd, vp, vp]),
        ("idocp_unparnmpc_compute_kkt_residual", [vp, cd, c_double_p, c_double_p]),
        ("idocp_unparnmpc_launch_phase", [vp, ci, vp, vp]),
        ("idocp_unparnmpc_get_new_solution", [vp, cs, ci, c_double_p]),
        ("idocp_unparnmpc_halo_size", [ci]),
        ("idocp_unparnmpc_export_halo", [vp, ci, vp]),
        ("idocp_unparnmpc_import_halo", [vp, ci, vp]),
        ("idocp_unparnmpc_prev_state", [vp, P(vp), P(vp)]),
        ("idocp_unparnmpc_step_sizes_device", [vp, P(vp)]),
        ("idocp_unparnmpc_kkt_error_squared_device", [vp, cd, vp]),
        ("idocp_unocp_launch_riccati", [vp, vp, vp]),
        ("idocp_unocp_launch_expand", [vp]),
        ("idocp_unocp_launch_integrate", [vp]),
        ("idocp_unocp_launch_kernel", [vp, ci, vp, vp]),
        ("idocp_rnea_derivatives", [P(Model), ci, c_double_p, c_double_p, c_double_p, c_double_p,
                                    c_double_p, c_double_p, c_double_p, ci]),
    ]:
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = ci
    lib.idocp_unocp_stream.argtypes = [vp]
    lib.idocp_unocp_stream.restype = vp
    lib.idocp_ocp_create.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, P(vp)]
    lib.idocp_ocp_create.restype = ci
    lib.idocp_parnmpc_create.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, P(vp)]
    lib.idocp_parnmpc_create.restype = ci
    lib.idocp_parnmpc_create_hybrid.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, ci, P(vp)]
    lib.idocp_parnmpc_create_hybrid.restype = ci
    lib.idocp_parnmpc_init_backward_correction.argtypes = [vp, cd]
    lib.idocp_parnmpc_init_backward_correction.restype = ci
    lib.idocp_parnmpc_update_solution.argtypes = [vp, cd, vp, vp, ci]
    lib.idocp_parnmpc_update_solution.restype = ci
    lib.idocp_parnmpc_compute_direction.argtypes = [vp, cd, vp, vp]
    lib.idocp_parnmpc_compute_direction.restype = ci
    lib.idocp_parnmpc_update_solution_device.argtypes = [vp, cd, vp, vp]
    lib.idocp_parnmpc_update_solution_device.restype = ci
    lib.idocp_parnmpc_launch_phase.argtypes = [vp, ci, vp, vp]
    lib.idocp_parnmpc_launch_phase.restype = ci
    lib.idocp_parnmpc_compute_kkt_residual.argtypes = [vp, cd, vp, vp]
    lib.idocp_parnmpc_compute_kkt_residual.restype = ci
    lib.idocp_parnmpc_create_shard.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, ci, ci, ci, P(vp)]
    lib.idocp_parnmpc_create_shard.restype = ci
    lib.idocp_parnmpc_create_hybrid_shard.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, ci, ci, ci, P(vp)]
    lib.idocp_parnmpc_create_hybrid_shard.restype = ci
    lib.idocp_parnmpc_halo_size.argtypes = [ci]
    lib.idocp_parnmpc_halo_size.restype = ci
    lib.idocp_parnmpc_export_halo.argtypes = [vp, ci, vp]
    lib.idocp_parnmpc_export_halo.restype = ci
    lib.idocp_parnmpc_import_halo.argtypes = [vp, ci, vp]
    lib.idocp_parnmpc_import_halo.restype = ci
    lib.idocp_parnmpc_prev_state.argtypes = [vp, P(vp), P(vp)]
    lib.idocp_parnmpc_prev_state.restype = ci
    lib.idocp_parnmpc_step_sizes_device.argtypes = [vp, P(vp)]
    lib.idocp_parnmpc_step_sizes_device.restype = ci
    lib.idocp_parnmpc_discretize.argtypes = [vp, cd]
    lib.idocp_parnmpc_discretize.restype = ci
    lib.idocp_parnmpc_kkt_error_squared_device.argtypes = [vp, cd, vp]
    lib.idocp_parnmpc_kkt_error_squared_device.restype = ci
    lib.idocp_device_copy.argtypes = [vp, vp, C.c_ulong]
    lib.idocp_device_copy.restype = ci
    lib.idocp_ocp_create_hybrid.argtypes = [P(Model), P(Cost), P(Constraints), cd, ci, ci, ci, ci, P(vp)]
    lib.idocp_ocp_create_hybrid.restype = ci
    lib.idocp_ocp_push_back_contact_status.argtypes = [vp, P(ci), vp, cd]
    lib.idocp_ocp_push_back_contact_status.restype = ci
    lib.idocp_ocp_set_contact_points.argtypes = [vp, ci, vp]
    lib.idocp_ocp_set_contact_points.restype = ci
    lib.idocp_ocp_pop_back_contact_status.argtypes = [vp]
    lib.idocp_ocp_pop_back_contact_status.restype = ci
    lib.idocp_ocp_pop_front_contact_status.argtypes = [vp]
    lib.idocp_ocp_pop_front_contact_status.restype = ci
    lib.idocp_ocp_get_chain.argtypes = [vp, cd, ci, P(ci), P(ci), P(ci), vp, P(ci), P(ci)]
    lib.idocp_ocp_get_chain.restype = ci
    lib.idocp_ocp_get_chain_times.argtypes = [vp, cd, ci, vp]
    lib.idocp_ocp_get_chain_times.restype = ci
    lib.idocp_ocp_set_task_refs.argtypes = [vp, cd, ci, vp]
    lib.idocp_ocp_set_task_refs.restype = ci
    lib.idocp_ocp_get_solution_chain.argtypes = [vp, cs, ci, vp]
    lib.idocp_ocp_get_solution_chain.restype = ci
    lib.idocp_ocp_get_direction_chain.argtypes = [vp, cs, ci, vp]
    lib.idocp_ocp_get_direction_chain.restype = ci
    lib.idocp_ocp_get_riccati_chain.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.idocp_ocp_get_riccati_chain.restype = ci
    lib.idocp_ocp_destroy.argtypes = [vp]
    lib.idocp_ocp_destroy.restype = None
    lib.idocp_ocp_stream.argtypes = [vp]
    lib.idocp_ocp_stream.restype = vp
    lib.idocp_comm_destroy.argtypes = [vp]
    lib.idocp_comm_destroy.restype = None
    for name, args in [
        ("idocp_ocp_set_contact_status_uniformly", [vp, P(ci), c_double_p]),
        ("idocp_ocp_set_solution", [vp, cs, c_double_p]),
        ("idocp_ocp_set_solution_batch", [vp, cs, c_double_p]),
        ("idocp_ocp_init_constraints", [vp, cd]),
        ("idocp_ocp_update_solution", [vp, cd, c_double_p, c_double_p, ci]),
        ("idocp_ocp_update_solution_device", [vp, cd, vp, vp]),
        ("idocp_ocp_update_solution_graph", [vp, cd, vp, vp]),
        ("idocp_ocp_set_solution_stages", [vp, C.c_char_p, ci, c_double_p]),
        ("idocp_parnmpc_set_aux_mat", [vp, ci, c_double_p]),
        ("idocp_ocp_set_solution_chain", [vp, C.c_char_p, ci, c_double_p]),
        ("idocp_parnmpc_set_aux_mat_chain", [vp, ci, c_double_p]),
        ("idocp_ocp_get_contact_dynamics", [vp, ci, ci, c_double_p, c_double_p, c_double_p]),
        ("idocp_ocp_get_contact_dynamics_chain", [vp, ci, ci, c_double_p, c_double_p, c_double_p]),
        ("idocp_ocp_set_lqr_stage", [vp, ci, ci] + [c_double_p] * 11),
        ("idocp_parnmpc_get_new_solution_chain", [vp, C.c_char_p, ci, c_double_p]),
        ("idocp_parnmpc_get_aux_mat_chain", [vp, ci, c_double_p]),
        ("idocp_comm_get_unique_id", [vp]),
        ("idocp_comm_init_rank", [vp, ci, ci, ci, C.POINTER(vp)]),
        ("idocp_comm_init_local", [ci, ci, C.POINTER(vp)]),
        ("idocp_comm_rank", [vp]),
        ("idocp_comm_set_force_collectives", [vp, ci]),
        ("idocp_parnmpc_dist_transport_selftest", [vp, c_double_p]),
        ("idocp_ocp_state_dims", [vp, C.POINTER(ci), C.POINTER(ci)]),
        ("idocp_comm_world", [vp]),
        ("idocp_parnmpc_dist_attach", [vp, vp]),
        ("idocp_parnmpc_dist_detach", [vp]),
        ("idocp_parnmpc_dist_set_initial_state", [vp, c_double_p, c_double_p, ci, ci]),
        ("idocp_parnmpc_dist_init_backward_correction", [vp, cd]),
        ("idocp_parnmpc_dist_update_solution", [vp, cd]),
        ("idocp_parnmpc_dist_update_solution_ls", [vp, cd]),
        ("idocp_parnmpc_dist_kkt_error", [vp, cd, c_double_p]),
        ("idocp_ocp_batch", [vp]),
        ("idocp_ocp_set_riccati_storage", [vp, ci]),
        ("idocp_ocp_clone", [vp, C.POINTER(vp)]),
        ("idocp_ocp_clear_line_search_filter", [vp]),
        ("idocp_ocp_compute_direction", [vp, cd, c_double_p, c_double_p]),
        ("idocp_ocp_line_search_eval", [vp, c_double_p, c_double_p, c_double_p]),
        ("idocp_ocp_synchronize", [vp]),
        ("idocp_ocp_compute_kkt_residual", [vp, cd, c_double_p, c_double_p]),
        ("idocp_ocp_kkt_error", [vp, c_double_p]),
        ("idocp_ocp_get_solution", [vp, cs, ci, c_double_p]),
        ("idocp_ocp_get_direction", [vp, cs, ci, c_double_p]),
        ("idocp_ocp_get_step_sizes", [vp, c_double_p, c_double_p]),
        ("idocp_ocp_get_riccati", [vp, ci, c_double_p, c_double_p, c_double_p, c_double_p]),
        ("idocp_ocp_get_state_feedback_gain", [vp, ci, ci, c_double_p, c_double_p]),
        ("idocp_ocp_dimc", [vp]),
        ("idocp_ocp_is_current_solution_feasible", [vp, c_int_p, c_int_p]),
        ("idocp_ocp_get_constraint_data", [vp, ci, c_double_p, c_double_p]),
        ("idocp_ocp_get_lqr_stage", [vp, ci, ci] + [c_double_p] * 8),
        ("idocp_ocp_launch_kernel", [vp, ci, vp, vp]),
        ("idocp_ocp_get_profile", [vp, C.POINTER(C.c_longlong), ci]),
    ]:
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = ci


def lib():
    """Load libidocp_hip.so (the HIP extension).  Fails loudly when it is missing:
    there is no CPU fallback for the product path."""
    global _lib, LIB_PATH
    if _lib is None:
        LIB_PATH = os.environ.get("IDOCP_HIP_LIB", LIB_PATH)      # diagnostics: another build of the same library (e.g. with clock stamps)
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                "HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _proto(_lib)
    return _lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def last_error():
    return lib().idocp_last_error().decode()


def check(rc, what="idocp call"):
    if rc != 0:
        raise RuntimeError("%s failed with status %d: %s" % (what, rc, last_error()))


def model_from_urdf(path, contact_frames=()):
    m = Model()
    cf = (C.c_int * max(1, len(contact_frames)))(*contact_frames)
    check(lib().idocp_model_from_urdf(path.encode(), cf, len(contact_frames), C.byref(m)), "idocp_model_from_urdf")
    return m
