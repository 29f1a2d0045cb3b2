"""GPU parity of JointAccelerationLowerLimit / JointAccelerationUpperLimit (SURVEY 8f row 3;
src/constraints/joint_acceleration_{lower,upper}_limit.cpp: a.tail(dimu) >= amin, <= amax with bounds of the components' own, active
on every stage with torques) against the oracle on OCPSolver -- uniform horizon and a trotting chain with impulse stages -- and on
ParNMPCSolver.  The bounds are tight enough to bind (the unconstrained first iterate exceeds them)."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC,
                     anymal_contact_points, anymal_model, anymal_problem, parity, rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu
TOL = 1e-10
A_MAX = 6.0


def problem(m, lower=True, upper=True, **kw):
    cost, cons = anymal_problem(m, **kw)
    cons.joint_acceleration_lower_limit = int(lower)
    cons.joint_acceleration_upper_limit = int(upper)
    for j in range(12):
        cons.a_min[j] = -A_MAX - 0.1 * j
        cons.a_max[j] = A_MAX + 0.05 * j
    return cost, cons


def start(solvers, m, seq=None):
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        if seq is None:
            s.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
        else:
            trotting_sequence(s, m, seq)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    return q, v


@pytest.mark.parametrize("lower,upper", [(True, True), (False, True), (True, False)], ids=["both", "upper", "lower"])
def test_ocp_uniform_horizon(lower, upper):
    m = anymal_model()
    cost, cons = problem(m, lower, upper, trotting_ref=False)
    o, g = OracleOCP(m, cost, cons, 0.5, 20), HipOCP(m, cost, cons, 0.5, 20)
    q, v = start((o, g), m)
    o.init_constraints(0.0); g.init_constraints(0.0)
    assert g.lib.idocp_ocp_dimc(g.h) == o.lib.oracle_ocp_dimc(o.h) == 6 * 12 + 5 * 4 + 12 * (int(lower) + int(upper))
    for a, b in zip(g.constraint_data(), o.constraint_data()):
        assert rel_err(a, b) < TOL                                     # setSlackAndDual (joint_acceleration_upper_limit.cpp:50-54)
    rng = np.random.default_rng(5)
    q[7:] += 0.005 * rng.uniform(-1, 1, 12)                            # the joints have to accelerate back: |a| of the free problem is 12 > A_MAX
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it == 0:
            for f in OCP_DIR_FIELDS:
                assert rel_err(g.get(f), o.get(f)) < TOL, f
            ao, bo = o.step_sizes()
            ag, bg = g.step_sizes()
            assert abs(ag[0] - ao) < 1e-10 and abs(bg[0] - bo) < 1e-10
            for f in OCP_SOL_FIELDS:
                assert rel_err(g.get(f), o.get(f)) < TOL, f
            for a, b in zip(g.constraint_data(), o.constraint_data()):
                assert rel_err(a, b) < TOL
    e_o2, e_g2 = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert e_g2 < 1e-6 * e_g and abs(np.log10(e_g2 / e_o2)) < 1.0
    a = g.get("a")[:, 6:]
    lo = np.array([cons.a_min[j] for j in range(12)]); hi = np.array([cons.a_max[j] for j in range(12)])
    if upper: assert (a <= hi + 1e-9).all() and (a > hi - 1e-2).any()      # feasible, and the bound binds somewhere
    if lower: assert (a >= lo - 1e-9).all()
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1]
    # accelerations beyond the bound are reported from the first stage on (joint_acceleration_{lower,upper}_limit.cpp:38-47)
    for s_ in (o, g):
        s_.set_solution("a", np.full(m.nv, 100.0 if upper else -100.0))
    assert o.infeasible_stage() == 0 and list(g.infeasible_stage()) == [0]


def test_ocp_trotting_chain_and_line_search():
    m = anymal_model()
    cost, cons = problem(m, trotting_ref=True)
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)      # long double referee
    g = HipOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    q, v = start((o, g, h), m, seq=nimp)
    o.init_constraints(0.0); g.init_constraints(0.0); h.init_constraints(0.0)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    # filter line search: the acceleration rows enter the barrier cost and the l1 violation (line_search.cpp:63-196)
    import ctypes as C
    from idocp_amd import capi
    from helpers import P
    for s_ in (o, h):
        s_.lib.oracle_ocp_update_solution_ls.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
        assert s_.lib.oracle_ocp_update_solution_ls(s_.h, 0.0, P(q), P(v)) == 0
    assert g.lib.idocp_ocp_update_solution(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv)), 1) == 0
    ao, _ = o.step_sizes()
    ag, _ = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-10
    # 1e-10 stage by stage; on the short stages around the impulse (1.7 ms on this grid) the long double referee decides
    for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
        parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), f, cap=1e-8)
    for f in ("q", "v", "a", "f", "u", "lmd", "gmm", "beta", "mu"):
        parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), f, cap=1e-8)


def test_parnmpc_horizon():
    m = anymal_model()
    cost, cons = problem(m, trotting_ref=False)
    o, g = OracleParNMPC(m, cost, cons, 0.5, 20), HipParNMPC(m, cost, cons, 0.5, 20)
    h = OracleParNMPC(m, cost, cons, 0.5, 20, hp=True)                         # long double referee
    q, v = start((o, g, h), m)
    o.init(0.0); g.init(0.0); h.init(0.0)
    rng = np.random.default_rng(5)
    q[7:] += 0.005 * rng.uniform(-1, 1, 12)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        # (cold-started ParNMPC: the correction sweeps amplify rounding along the horizon, tests/test_parnmpc_gpu.py: the referee decides)
        parity(g.get(f), o.get(f), lambda f=f: h.get(f), f, cap=5e-8)
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-8 and abs(bg[0] - bo) < 1e-8
