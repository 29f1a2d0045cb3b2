"""GPU parity of the ParNMPC path (backward-Euler stages, per-stage KKT inverse, backward correction) against the
oracle, event-free horizon with 4 active point contacts (examples/anymal/parnmpc_benchmark.cpp shape).
Bar: 1e-10 on the Newton direction of the first iteration (FP64)."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipParNMPC, OracleParNMPC, anymal_contact_points,
                     anymal_model, anymal_problem, rel_err)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def make_pair(N, T, batch=1, trotting_ref=False):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=trotting_ref)
    o = OracleParNMPC(m, cost, cons, T, N)
    g = HipParNMPC(m, cost, cons, T, N, batch=batch)
    pts = anymal_contact_points(m)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        s.set_contact_status([1, 1, 1, 1], pts)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init(0.0)
    g.init(0.0)
    qq = q.copy()
    qq[7:] += 0.05
    return m, o, g, qq, v


@pytest.mark.parametrize("N,T", [(20, 0.5), (64, 3.2)])
def test_first_iteration_direction_parity(N, T):
    m, o, g, q, v = make_pair(N, T)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0
    assert g.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-10 and abs(bg[0] - bo) < 1e-10
    # the updated iterate s + alpha d: same bar on the short horizon; the 84 x 84 KKT inverses of the long one (Gauss-Jordan
    # here, two LLTs in the oracle) leave 1.1e-10 on u
    for f in OCP_SOL_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < (TOL if N <= 20 else 1e-9), f


def test_convergence_and_batch():
    m, o, g, q, v = make_pair(20, 0.5, batch=3)
    for it in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
        assert np.allclose(e_g, e_g[0], rtol=0, atol=1e-9 * max(1.0, e_o))
        assert abs(e_g[0] - e_o) <= (1e-10 if it == 0 else 1e-6) * max(1.0, e_o) + 1e-10, (it, e_g[0], e_o)
    assert e_g[0] < 1e-8
    for f in ("q", "v", "a", "u", "f"):
        assert rel_err(g.get(f, 2), o.get(f)) < 1e-6, f
