"""GPU parity of ContactDistance (src/constraints/contact_distance.cpp: the frames of the contacts that are not active
stay above z = 0; rows of the LOCAL frame Jacobian, evaluated outside the condensation kernel by ocp_ext_kernel.hip) against the oracle:
OCPSolver on a trotting chain (swing feet, impulse stages), its filter line search, ParNMPCSolver on a chain with a lift."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, P, anymal_contact_points,
                     anymal_model, anymal_problem, parity, rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu


def ocp_pair(batch=1, referee=False):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    cons.contact_distance = 1
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    solvers = [o, g]
    if referee:
        solvers.append(OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True))      # long double referee
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        trotting_sequence(s, m, nimp)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    return (m, o, g, q, v, solvers[2]) if referee else (m, o, g, q, v)


def test_ocp_trotting_chain():
    m, o, g, q, v, h = ocp_pair(referee=True)
    assert g.lib.idocp_ocp_dimc(g.h) == o.lib.oracle_ocp_dimc(o.h) == 6 * 12 + 5 * 4 + 4
    for a, b in zip(g.constraint_data(), o.constraint_data()):
        assert rel_err(a, b) < 1e-10                                   # setSlackAndDual: heights of the frames (contact_distance.cpp:58-65)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(12):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it <= 3:
            assert h.update(0.0, q, v) == 0
        if it in (0, 3):
            # rows with slack ~ 1e-4 weigh 1e4 in the Hessian: where two FP64 evaluations separate beyond 1e-10 the long double referee
            # decides (GPU at most 4x as far from it as the FP64 oracle, + 1e-10); 2e-8 is only the cap against the oracle
            for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
                parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), (it, f), cap=2e-8)
            tol = 1e-9 if it == 0 else 2e-8
            ao, bo = o.step_sizes()
            ag, bg = g.step_sizes()
            assert abs(ag[0] - ao) < tol and abs(bg[0] - bo) < tol
            for a, b, c in zip(g.constraint_data(), o.constraint_data(), h.constraint_data()):
                parity(a, b, c, ("slack / dual", it), cap=2e-8)
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
    h = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3, hp=True)      # long double referee
    g = HipParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g, h):
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
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        parity(g.get_chain(f, M + 1)[:M], o.get_chain(f, M), lambda f=f: h.get_chain(f, M), f, cap=1e-8)


def test_clone_and_hipgraph_carry_the_ext_kernels():
    """The kernels around the condensation (ocp_ext_kernel.hip) are part of what idocp_ocp_clone copies (their records are allocations of
    the handle) and of what idocp_ocp_update_solution_graph captures: a clone continues like the original, and the iteration replayed
    from the graph gives the iterate of the eager launches, with ContactDistance AND a task-space cost switched on."""
    import copy
    import torch
    from idocp_amd.workloads import ANYMAL_URDF
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    cons.contact_distance = 1
    lib = capi.lib()
    fid = lib.idocp_model_frame_id(ANYMAL_URDF.encode(), b"base")
    joint = C.c_int()
    R, p = (C.c_double * 9)(), (C.c_double * 3)()
    capi.check(lib.idocp_model_frame_placement(ANYMAL_URDF.encode(), fid, C.byref(joint), R, p), "frame_placement")
    cost.task_dim, cost.task_joint = 3, joint.value
    for k in range(9):
        cost.task_frame_R[k] = R[k]
        cost.task_ref[k] = 1.0 if k % 4 == 0 else 0.0
    for k, x in enumerate((0.05, 0.0, 0.5)):
        cost.task_frame_p[k] = p[k]
        cost.task_ref[9 + k] = x
        cost.task_weight[k] = cost.task_weightf[k] = cost.task_weighti[k] = 50.0
    N, T, nimp = 31, 1.55, 2

    def make():
        g = HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=nimp + 1)
        trotting_sequence(g, m, nimp)
        g.set_solution("q", ANYMAL_Q_STANDING)
        g.set_solution("v", np.zeros(m.nv))
        g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        g.init_constraints(0.0)
        return g
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    g = make()
    for _ in range(2):
        assert g.update(0.0, q, v) == 0
    h2 = C.c_void_p()
    capi.check(lib.idocp_ocp_clone(g.h, C.byref(h2)), "clone")
    c = copy.copy(g)
    c.h = h2
    for s_ in (g, c):
        for _ in range(2):
            assert s_.update(0.0, q, v) == 0
    M = len(g.chain(0.0))
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm"):
        assert np.array_equal(g.get_chain(f, M, instance=1), c.get_chain(f, M, instance=1)), f
    # eager vs graph replay
    e, r = make(), make()
    dq = torch.tensor(np.tile(q, (2, 1)), dtype=torch.float64, device="cuda")
    dv = torch.tensor(np.tile(v, (2, 1)), dtype=torch.float64, device="cuda")
    for _ in range(3):
        capi.check(lib.idocp_ocp_update_solution_device(e.h, 0.0, C.c_void_p(dq.data_ptr()), C.c_void_p(dv.data_ptr())), "eager")
        capi.check(lib.idocp_ocp_update_solution_graph(r.h, 0.0, C.c_void_p(dq.data_ptr()), C.c_void_p(dv.data_ptr())), "graph")
    capi.check(lib.idocp_ocp_synchronize(e.h)); capi.check(lib.idocp_ocp_synchronize(r.h))
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm"):
        assert np.array_equal(e.get_chain(f, M), r.get_chain(f, M)), f


def test_slack_initialisation_far_below_the_ground_stays_positive_and_finite():
    # pdipm::SetSlackAndDualPositive (pdipm.hxx:13-23) adds the barrier until the slack reaches it.  With the base a metre under the
    # ground the rows start 1e4 barriers below: the device lifts them in one step behind the first 1024 additions (slackPositive,
    # unocp_device.hpp) -- the slacks must come out >= barrier, finite, with dual = barrier / slack, and the first Newton step must be
    # free of NaN (round 3 capped the loop and left such a slack negative: negative dual, log of a negative number in the barrier cost).
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    cons.contact_distance = 1
    N, T, nimp = 31, 1.55, 2
    g = HipOCP(m, cost, cons, T, N, batch=1, max_num_impulse=nimp + 1)
    trotting_sequence(g, m, nimp)
    q = ANYMAL_Q_STANDING.copy()
    q[2] -= 1.0
    g.set_solution("q", q)
    g.set_solution("v", np.zeros(m.nv))
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    slack, dual = g.constraint_data()
    slack, dual = np.asarray(slack), np.asarray(dual)
    assert np.all(np.isfinite(slack)) and np.all(np.isfinite(dual))
    rows = slack[:, -4:] if slack.ndim == 2 else slack.reshape(-1, slack.shape[-1])[:, -4:]
    drows = dual[:, -4:] if dual.ndim == 2 else dual.reshape(-1, dual.shape[-1])[:, -4:]
    live = drows > 0                                        # rows that exist (stages with level >= 2 that are not impulse stages)
    assert live.any()
    assert np.all(rows[live] >= cons.barrier * (1 - 1e-12))
    assert np.all(rows[live] < cons.barrier * 3)            # lifted just above the barrier, like the reference's loop would
    assert np.allclose(drows[live], cons.barrier / rows[live], rtol=1e-14)
