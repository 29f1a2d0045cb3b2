"""Parity on RANDOM problems: horizon, time step, the feet in contact (none, one, two, three, all four -- every stage class of the condensation kernel,
the odd contact counts included), cost weights over four decades, the friction coefficient, the start state.  The hand-written cases of the other test
files sit at the corners someone thought of; this one draws from the space between them.  Bar: 1e-10 on the first Newton direction against the oracle,
stage by stage, or the long double referee's word where two FP64 evaluation orders cannot agree that far (helpers.parity)."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem,
                     parity)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def random_problem(rng, m):
    cost, cons = anymal_problem(m, trotting_ref=False)
    nv = m.nv
    lw = lambda lo, hi, n: 10.0 ** rng.uniform(lo, hi, size=n)
    cost.set("q_weight", lw(-1, 2, nv)).set("qf_weight", lw(-1, 2, nv)).set("v_weight", lw(-2, 1, nv)).set("vf_weight", lw(-2, 1, nv))
    cost.set("a_weight", lw(-4, -1, nv)).set("u_weight", lw(-5, -2, nv))
    for c in range(4):
        for k in range(3):
            cost.f_weight[c][k] = float(10.0 ** rng.uniform(-4, -2))
            cost.f_ref[c][k] = float(rng.uniform(-5, 5)) if k < 2 else float(rng.uniform(20, 120))
    cons.mu = float(rng.uniform(0.4, 1.0))
    return cost, cons


@pytest.mark.parametrize("seed", range(12))
def test_first_direction_on_random_problems(seed):
    rng = np.random.default_rng(1000 + seed)
    m = anymal_model()
    cost, cons = random_problem(rng, m)
    N = int(rng.integers(3, 25))
    T = N * float(rng.uniform(0.01, 0.05))
    active = [int(x) for x in rng.integers(0, 2, size=4)]
    if seed < 5:
        active = [[0, 0, 0, 0], [1, 0, 0, 0], [1, 0, 1, 0], [0, 1, 1, 1], [1, 1, 1, 1]][seed]      # every contact count at least once
    pts = anymal_contact_points(m)
    q = ANYMAL_Q_STANDING.copy()
    q[:3] += rng.uniform(-0.05, 0.05, 3)
    quat = q[3:7] + rng.uniform(-0.05, 0.05, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.uniform(-0.15, 0.15, 12)
    v = rng.uniform(-0.3, 0.3, m.nv)
    fz = [float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), float(rng.uniform(40, 110))]
    par = seed % 3 == 2                                       # every third problem through ParNMPCSolver
    Hip, Orc = (HipParNMPC, OracleParNMPC) if par else (HipOCP, OracleOCP)
    solvers = [Hip(m, cost, cons, T, N, batch=2), Orc(m, cost, cons, T, N), Orc(m, cost, cons, T, N, hp=True)]
    for s in solvers:
        s.set_contact_status(active, pts)
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", fz)
        s.init(0.0) if par else s.init_constraints(0.0)
    g, o, h = solvers
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    ran_referee = []

    def referee(name):
        if not ran_referee:
            assert h.update(0.0, q, v) == 0
            ran_referee.append(1)
        return h.get(name)

    worst = 0.0
    for name in OCP_DIR_FIELDS:
        worst = max(worst, parity(g.get(name, 1), o.get(name), lambda name=name: referee(name), (seed, active, N, name), tol=TOL, cap=1e-7))
    print("seed %d  contacts %s  N %d  %s  worst %.2e%s" % (seed, active, N, "ParNMPC" if par else "OCP", worst, "  (referee consulted)" if ran_referee else ""))
