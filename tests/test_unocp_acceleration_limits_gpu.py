"""GPU parity of JointAccelerationLowerLimit / UpperLimit on the fixed-base solvers (UnOCPSolver, UnParNMPCSolver; iiwa14): rows of
their own (slack_a / dual_a, allocated only when a component is in use) next to the six joint-limit families."""
import numpy as np
import pytest

from helpers import HipUnOCP, HipUnParNMPC, OracleUnOCP, OracleUnParNMPC, iiwa14_model, rel_err, unocp_problem

pytestmark = pytest.mark.gpu
TOL = 1e-10
SOL = ("q", "v", "a", "u", "lmd", "gmm", "beta")


def problem(m, lower=1, upper=1, amax=4.0):
    cost, cons = unocp_problem(m)
    cons.joint_acceleration_lower_limit = lower
    cons.joint_acceleration_upper_limit = upper
    for j in range(m.nv):
        cons.a_min[j] = -amax - 0.1 * j
        cons.a_max[j] = amax + 0.05 * j
    return cost, cons


@pytest.mark.parametrize("lower,upper", [(1, 1), (0, 1), (1, 0)], ids=["both", "upper", "lower"])
def test_unocp_direction_iterate_and_convergence(lower, upper):
    m = iiwa14_model()
    cost, cons = problem(m, lower, upper)
    o, g = OracleUnOCP(m, cost, cons, 1.0, 20), HipUnOCP(m, cost, cons, 1.0, 20, batch=2)
    q, v = np.full(m.nv, 0.4), np.zeros(m.nv)
    for s in (o, g):
        s.set_solution("q", q)
        s.set_solution("v", v)
    assert g.lib.idocp_unocp_dimc(g.h) == o.lib.oracle_unocp_dimc(o.h) == 6 * m.nv + m.nv * (lower + upper)
    for a, b in zip(g.constraint_data(1), o.constraint_data()):
        assert rel_err(a, b) < TOL                          # setSlackAndDual at creation (a = 0: slack = the bound)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[1] - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it in (0, 3):
            tol = TOL if it == 0 else 1e-8
            for f in SOL:
                assert rel_err(g.direction("d" + f, 1), o.direction("d" + f)) < tol, (it, "d" + f)
                assert rel_err(g.solution(f, 1), o.solution(f)) < tol, (it, f)
            for a, b in zip(g.constraint_data(1), o.constraint_data()):
                assert rel_err(a, b) < tol
    e_o2, e_g2 = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g2[1] - e_o2) < 1e-6 * max(1.0, e_o2)      # (bounds this tight make the interior-point iteration crawl -- in both alike)
    a = g.solution("a", 1)
    lo = np.array([cons.a_min[j] for j in range(m.nv)]); hi = np.array([cons.a_max[j] for j in range(m.nv)])
    if upper: assert (a <= hi + 1e-9).all()
    if lower: assert (a >= lo - 1e-9).all()
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]
    bad = np.full(m.nv, 100.0 if upper else -100.0)
    o.set_solution("a", bad); g.set_solution("a", bad)
    assert o.infeasible_stage() == 0 and list(g.infeasible_stage()) == [0, 0]


def test_the_bound_binds():
    m = iiwa14_model()
    cost, cons = problem(m, amax=1e3)
    g = HipUnOCP(m, cost, cons, 1.0, 20)
    q, v = np.full(m.nv, 0.4), np.zeros(m.nv)
    g.set_solution("q", q); g.set_solution("v", v)
    for _ in range(30):
        assert g.update(0.0, q, v) == 0
    a_free = np.abs(g.solution("a")).max()
    cost, cons = problem(m, amax=0.5 * a_free)
    g2 = HipUnOCP(m, cost, cons, 1.0, 20)
    g2.set_solution("q", q); g2.set_solution("v", v)
    for _ in range(40):
        assert g2.update(0.0, q, v) == 0
    a = np.abs(g2.solution("a"))
    assert a.max() <= 0.5 * a_free + 0.1 * (m.nv - 1) + 1e-9 and a.max() > 0.45 * a_free      # (the lower bounds are - amax - 0.1 j)


def test_unocp_line_search_and_unparnmpc():
    m = iiwa14_model()
    cost, cons = problem(m, amax=60.0)          # (about half of what the free problem asks for: binding, and well posed)
    o, g = OracleUnOCP(m, cost, cons, 1.0, 20), HipUnOCP(m, cost, cons, 1.0, 20)
    q, v = np.full(m.nv, 0.4), np.zeros(m.nv)
    for s in (o, g):
        s.set_solution("q", q)
        s.set_solution("v", v)
    for it in range(4):
        assert o.update(0.0, q, v, line_search=True) == 0 and g.update(0.0, q, v, line_search=True) == 0
        ao, _ = o.step_sizes()
        ag, _ = g.step_sizes()
        assert abs(ag[0] - ao) < 1e-10, (it, ag[0], ao)
        for f in SOL:
            assert rel_err(g.solution(f), o.solution(f)) < 1e-8, (it, f)
    op, gp = OracleUnParNMPC(m, cost, cons, 1.0, 20), HipUnParNMPC(m, cost, cons, 1.0, 20)
    for s in (op, gp):
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.init(0.0)
    e_o, e_g = op.kkt_error(0.0, q, v), gp.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert op.update(0.0, q, v) == 0 and gp.update(0.0, q, v) == 0
    for f in ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta"):
        assert rel_err(gp.get(f), op.get(f)) < 1e-9, f
