"""CPU checks of the ParNMPC half of the oracle (event-free horizons): the backward-correction iteration of
src/ocp/parnmpc_solver.cpp:73-103 on the problem of examples/anymal/parnmpc_benchmark.cpp (N = 20, T = 0.5,
4 active point contacts).  The HIP ParNMPC path is not built yet; this oracle is
its future checker and already pins the stage linearisation shared with the OCPSolver path."""
import numpy as np

from helpers import ANYMAL_Q_STANDING, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem


def make(N=20, T=0.5):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    p = OracleParNMPC(m, cost, cons, T, N)
    p.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    p.set_solution("q", q)
    p.set_solution("v", v)
    p.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    p.init(0.0)
    return m, cost, cons, p, q, v


def test_parnmpc_converges_to_a_kkt_point():
    m, cost, cons, p, q, v = make()
    qq = q.copy()
    qq[7:] += 0.05
    errs = []
    for it in range(35):                       # ocpbenchmarker::Convergence(parnmpc_solver, t, q, v, 20, false) and a few more
        assert p.update(0.0, qq, v) == 0
        errs.append(p.kkt_error(0.0, qq, v))
    assert np.isfinite(errs).all() and errs[-1] < 1e-9
    # the backward-Euler chain is satisfied: q_prev (-) q_i + dt v_i = 0 on the joints, v_prev - v_i + dt a_i = 0
    Q, V, A = p.get("q"), p.get("v"), p.get("a")
    dt = 0.5 / 20
    prev_q, prev_v = qq, v
    for i in range(20):
        assert np.abs(prev_q[7:] - Q[i, 7:] + dt * V[i, 6:]).max() < 1e-9
        assert np.abs(prev_v - V[i] + dt * A[i]).max() < 1e-9
        prev_q, prev_v = Q[i], V[i]
