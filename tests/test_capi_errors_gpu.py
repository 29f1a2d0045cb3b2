"""Error behaviour of the C ABI on a live handle: where the reference throws and exits (std::out_of_range /
std::invalid_argument / std::runtime_error followed by std::exit(EXIT_FAILURE), e.g. src/ocp/ocp_solver.cpp:20-57, 95-165,
include/idocp/hybrid/contact_sequence.hxx:63-117) the library returns IDOCP_E_ARG and leaves the reference's message in
idocp_last_error(); nothing is silently accepted, nothing falls back to a CPU path."""
import ctypes as C

import numpy as np
import pytest

from helpers import ANYMAL_Q_STANDING, HipOCP, HipParNMPC, P, anymal_contact_points, anymal_model, anymal_problem, arr
from idocp_amd import capi

pytestmark = pytest.mark.gpu
E_ARG, E_UNSUPPORTED = -1, -4


def test_contact_sequence_errors():
    m = anymal_model()
    cost, cons = anymal_problem(m)
    lib = capi.lib()
    g = HipOCP(m, cost, cons, 1.0, 20, max_num_impulse=2)
    pts = anymal_contact_points(m)
    a = lambda s: (C.c_int * 4)(*s)
    # push_back before setContactStatusUniformly
    assert lib.idocp_ocp_push_back_contact_status(g.h, a([0, 1, 1, 0]), P(arr(pts)), 0.3) == E_ARG
    assert "setContactStatusUniformly" in capi.last_error()
    g.set_contact_status([1, 1, 1, 1], pts)
    # a status equal to the last one is no discrete event
    assert lib.idocp_ocp_push_back_contact_status(g.h, a([1, 1, 1, 1]), P(arr(pts)), 0.3) == E_ARG
    assert "existDiscreteEvent" in capi.last_error()
    g.push_back_contact_status([0, 1, 1, 0], pts, 0.3)
    # event times must increase
    assert lib.idocp_ocp_push_back_contact_status(g.h, a([1, 1, 1, 1]), P(arr(pts)), 0.3) == E_ARG
    assert "must be larger than the last event time" in capi.last_error()
    g.push_back_contact_status([1, 1, 1, 1], pts, 0.5)          # impulse 1
    g.push_back_contact_status([0, 1, 1, 0], pts, 0.6)          # lift 2
    g.push_back_contact_status([1, 1, 1, 1], pts, 0.7)          # impulse 2
    # a third lift does not fit max_num_impulse = 2
    assert lib.idocp_ocp_push_back_contact_status(g.h, a([0, 1, 1, 0]), P(arr(pts)), 0.8) == E_ARG
    assert "max_num_impulse" in capi.last_error()
    # contact phase out of range
    assert lib.idocp_ocp_set_contact_points(g.h, 9, P(arr(pts))) == E_ARG
    # two events inside one interval of the grid are rejected by the discretiser when the horizon is used
    g2 = HipOCP(m, cost, cons, 1.0, 20, max_num_impulse=2)
    g2.set_contact_status([1, 1, 1, 1], pts)
    g2.push_back_contact_status([0, 1, 1, 0], pts, 0.31)
    g2.push_back_contact_status([1, 1, 1, 1], pts, 0.33)
    assert lib.idocp_ocp_init_constraints(g2.h, 0.0) == E_ARG
    assert "same time stage" in capi.last_error()


def test_solver_argument_errors():
    m = anymal_model()
    cost, cons = anymal_problem(m)
    lib = capi.lib()
    h = C.c_void_p()
    for T, N, nimp, batch in ((0.0, 20, 0, 1), (1.0, 0, 0, 1), (1.0, 20, -1, 1), (1.0, 20, 0, 0)):
        assert lib.idocp_ocp_create_hybrid(C.byref(m), C.byref(cost), C.byref(cons), T, N, nimp, batch, 0, C.byref(h)) == E_ARG
    g = HipOCP(m, cost, cons, 1.0, 20)
    assert lib.idocp_ocp_set_solution(g.h, b"w", P(np.zeros(19))) == E_ARG
    assert "name must be q, v, a, f, or u" in capi.last_error()
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    # updateSolution before a contact status was given (with and without the line search)
    assert lib.idocp_ocp_update_solution(g.h, 0.0, P(q), P(v), 0) == E_ARG
    assert lib.idocp_ocp_update_solution(g.h, 0.0, P(q), P(v), 1) == E_ARG
    g.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    # getters: unknown field, instance out of range
    out = np.zeros((21, 19))
    assert lib.idocp_ocp_get_solution(g.h, b"nope", 0, P(out)) == E_ARG
    assert lib.idocp_ocp_get_solution(g.h, b"q", 5, P(out)) == E_ARG


def test_parnmpc_unsupported_cases_fail_loudly():
    m = anymal_model()
    cost, cons = anymal_problem(m)
    lib = capi.lib()
    pts = anymal_contact_points(m)
    # discrete events on a handle created without room for them
    g = HipParNMPC(m, cost, cons, 1.0, 20)
    g.set_contact_status([1, 1, 1, 1], pts)
    a = (C.c_int * 4)(0, 1, 1, 0)
    assert lib.idocp_ocp_push_back_contact_status(g.h, a, P(arr(pts)), 0.52) == E_ARG
    # an impulse inside the first interval of a ParNMPC horizon is carried (tests/test_parnmpc_hybrid_gpu.py, "impulse-first")
    g2 = HipParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=2)
    g2.set_contact_status([0, 1, 1, 0], pts)
    g2.push_back_contact_status([1, 1, 1, 1], pts, 0.02)
    assert lib.idocp_parnmpc_init_backward_correction(g2.h, 0.0) == 0


def test_fixed_base_solver_argument_errors():
    """UnOCPSolver / UnParNMPCSolver constructor checks (unocp_solver.cpp:33-47, unparnmpc_solver.cpp:11-44), the shard range,
    and the entry points that belong to the other solver kind."""
    import ctypes as C
    from helpers import HipUnOCP, HipUnParNMPC, iiwa14_model, unocp_problem
    from idocp_amd import capi
    lib = capi.lib()
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    h = C.c_void_p()
    for fn in (lib.idocp_unocp_create, lib.idocp_unparnmpc_create):
        assert fn(C.byref(m), C.byref(cost), C.byref(cons), -1.0, 10, 1, 0, C.byref(h)) == E_ARG
        assert b"T must be positive" in lib.idocp_last_error()
        assert fn(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 0, 1, 0, C.byref(h)) == E_ARG
        assert b"N must be positive" in lib.idocp_last_error()
    # the components the fixed-base solvers do not carry are refused, not dropped
    cons.contact_distance = 1          # (the acceleration limits are carried: tests/test_unocp_acceleration_limits_gpu.py)
    for fn in (lib.idocp_unocp_create, lib.idocp_unparnmpc_create):
        assert fn(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, 1, 0, C.byref(h)) == E_UNSUPPORTED
    cons.contact_distance = 0
    am = anymal_model()
    assert lib.idocp_unparnmpc_create(C.byref(am), C.byref(cost), C.byref(cons), 1.0, 10, 1, 0, C.byref(h)) == E_ARG     # floating base
    for lo, hi in ((-1, 5), (5, 5), (8, 11)):
        assert lib.idocp_unparnmpc_create_shard(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, lo, hi, 1, 0, C.byref(h)) == E_ARG
    g, u = HipUnParNMPC(m, cost, cons, 1.0, 4), HipUnOCP(m, cost, cons, 1.0, 4)
    d = C.c_void_p()
    capi.check(lib.idocp_device_alloc(C.byref(d), 8 * 64))
    assert lib.idocp_unparnmpc_export_halo(u.h, 0, d) == E_ARG and lib.idocp_unparnmpc_export_halo(g.h, 7, d) == E_ARG
    assert lib.idocp_unparnmpc_init_backward_correction(u.h, 0.0) == E_ARG
    assert lib.idocp_unocp_launch_kernel(g.h, 0, d, d) == E_ARG
    assert lib.idocp_unparnmpc_launch_phase(g.h, 9, d, d) == E_ARG
    # a horizon shard has no line search
    sh = C.c_void_p()
    capi.check(lib.idocp_unparnmpc_create_shard(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, 0, 5, 1, 0, C.byref(sh)))
    q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
    assert lib.idocp_unparnmpc_update_solution(sh, 0.0, P(arr(q)), P(arr(v)), 1) == E_UNSUPPORTED
    lib.idocp_unocp_destroy(sh)


def test_interior_point_parameters_are_validated():
    """A zeroed idocp_constraints_t (barrier = 0) or a rate outside (0, 1] is rejected at create
    (ConstraintComponentBase::setBarrier / setFractionToBoundaryRate assert the same, constraint_component_base.hxx:10-24):
    on the device `while (slack < barrier) slack += barrier` (pdipm.hxx:17-20) would never end."""
    import copy
    from helpers import iiwa14_model, unocp_problem
    lib = capi.lib()
    h = C.c_void_p()
    m = anymal_model()
    cost, cons = anymal_problem(m)
    for barrier, rate in ((0.0, 0.995), (-1e-4, 0.995), (1e-4, 0.0), (1e-4, 1.5), (float("nan"), 0.995)):
        bad = copy.copy(cons)
        bad.barrier, bad.fraction_to_boundary_rate = barrier, rate
        assert lib.idocp_ocp_create_hybrid(C.byref(m), C.byref(cost), C.byref(bad), 1.0, 20, 0, 1, 0, C.byref(h)) == E_ARG
        assert ("barrier" in capi.last_error()) or ("fraction_to_boundary_rate" in capi.last_error())
    mi = iiwa14_model()
    costi, consi = unocp_problem(mi)
    for fn in (lib.idocp_unocp_create, lib.idocp_unparnmpc_create):
        bad = copy.copy(consi)
        bad.barrier = 0.0
        assert fn(C.byref(mi), C.byref(costi), C.byref(bad), 1.0, 10, 1, 0, C.byref(h)) == E_ARG
        assert "barrier" in capi.last_error()
