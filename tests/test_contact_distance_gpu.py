"""GPU parity of ContactDistance (src/constraints/contact_distance.cpp: the frames of the contacts that are not active
stay above z = 0; rows of the LOCAL frame Jacobian, evaluated outside the condensation kernel by ocp_ext_kernel.hip) against the oracle:
OCPSolver on a trotting chain (swing feet, impulse stages), its filter line search, ParNMPCSolver on a chain with a lift."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, P, anymal_contact_points,
                     anymal_model, anymal_problem, rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu


def ocp_pair(batch=1):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    cons.contact_distance = 1
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        trotting_sequence(s, m, nimp)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    return m, o, g, q, v


def test_ocp_trotting_chain():
    m, o, g, q, v = ocp_pair()
    assert g.lib.idocp_ocp_dimc(g.h) == o.lib.oracle_ocp_dimc(o.h) == 6 * 12 + 5 * 4 + 4
    for a, b in zip(g.constraint_data(), o.constraint_data()):
        assert rel_err(a, b) < 1e-10                                   # setSlackAndDual: heights of the frames (contact_distance.cpp:58-65)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(12):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it in (0, 3):
            # (rows with slack ~ 1e-4 weigh 1e4 in the Hessian: the iterates of two FP64 evaluations separate a little faster than elsewhere)
            tol = 1e-9 if it == 0 else 2e-8
            for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
                assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < tol, (it, f)
            ao, bo = o.step_sizes()
            ag, bg = g.step_sizes()
            assert abs(ag[0] - ao) < tol and abs(bg[0] - bo) < tol
            for a, b in zip(g.constraint_data(), o.constraint_data()):
                assert rel_err(a, b) < tol, it
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-6 * max(1.0, e_o)
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1]
    # a configuration with the swing feet below the ground is reported on the first stage that carries the rows
    qb = ANYMAL_Q_STANDING.copy()
    qb[2] -= 0.1
    for s_ in (o, g):
        s_.set_solution("q", qb)
    so, sg = o.infeasible_stage(), list(g.infeasible_stage())[0]
    assert so == sg and so >= 2


def test_ocp_line_search_cost_and_violation():
    m, o, g, q, v = ocp_pair(batch=2)
    o.lib.oracle_ocp_cost_and_violation.argtypes = [C.c_void_p, C.c_double, capi.c_double_p]
    o.lib.oracle_ocp_compute_direction.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
    q[7:] += 0.05
    assert o.lib.oracle_ocp_compute_direction(o.h, 0.0, P(q), P(v)) == 0
    capi.check(g.lib.idocp_ocp_compute_direction(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))), "compute_direction")
    ap, _ = g.step_sizes()
    for alpha in (0.0, 0.01 * ap[0], 0.5 * ap[0], ap[0]):
        ref = np.zeros(2)
        assert o.lib.oracle_ocp_cost_and_violation(o.h, alpha, P(ref)) == 0
        c, vi = np.zeros(g.batch), np.zeros(g.batch)
        capi.check(g.lib.idocp_ocp_line_search_eval(g.h, P(np.full(g.batch, alpha)), P(c), P(vi)), "line_search_eval")
        assert abs(c[0] - ref[0]) <= 1e-9 * max(1.0, abs(ref[0])), (alpha, c[0], ref[0])
        assert abs(vi[0] - ref[1]) <= 1e-9 * max(1.0, abs(ref[1])), (alpha, vi[0], ref[1])
        assert c[0] == c[-1] and vi[0] == vi[-1]


def test_parnmpc_chain_with_a_lift():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    cons.contact_distance = 1
    o = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
    g = HipParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        pts = anymal_contact_points(m).copy()
        s.set_contact_status([1, 1, 1, 1], pts)
        s.push_back_contact_status([0, 1, 1, 0], pts, 0.52)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init(0.0)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) <= 1e-9 * e_o
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        assert rel_err(g.get_chain(f, M + 1)[:M], o.get_chain(f, M)) < 1e-9, f
