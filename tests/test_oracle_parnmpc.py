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


def test_line_search_merit_of_the_event_free_horizon():
    """ParNMPC half of LineSearch::computeCostAndViolation (src/line_search/line_search.cpp:199-237): at alpha = 0 the violation is the
    l1 norm of the residuals the KKT error squares (state equation, [ID - u; C], IPM primal residuals), the converged point has none
    left, and updateSolution(t, q, v, true) converges with accepted steps no larger than the fraction-to-boundary step."""
    from helpers import P
    m, cost, cons, p, q, v = make()
    qq = q.copy()
    qq[7:] += 0.05
    assert p.lib.oracle_parnmpc_compute_direction(p.h, 0.0, P(qq), P(v)) == 0
    cv0, cv1 = np.zeros(2), np.zeros(2)
    assert p.lib.oracle_parnmpc_cost_and_violation(p.h, 0.0, P(qq), P(v), P(cv0)) == 0
    amax, _ = p.step_sizes()
    assert p.lib.oracle_parnmpc_cost_and_violation(p.h, amax, P(qq), P(v), P(cv1)) == 0
    assert cv0[1] > 1e-3 and np.isfinite(cv0).all() and np.isfinite(cv1).all()
    # (no monotonicity along the step: the reference measures the IPM residual with the CURRENT slack, and the first ParNMPC
    #  direction from a cold start is not a Newton direction of the whole horizon)
    m, cost, cons, p, q, v = make()
    steps = []
    e0 = p.kkt_error(0.0, qq, v)
    for it in range(60):
        assert p.lib.oracle_parnmpc_update_solution_ls(p.h, 0.0, P(qq), P(v)) == 0
        steps.append(p.step_sizes()[0])
    assert p.kkt_error(0.0, qq, v) < 1e-6 * e0
    assert min(steps) >= 0.05 and max(steps) <= 1.0
    cv = np.zeros(2)
    assert p.lib.oracle_parnmpc_compute_direction(p.h, 0.0, P(qq), P(v)) == 0
    assert p.lib.oracle_parnmpc_cost_and_violation(p.h, 0.0, P(qq), P(v), P(cv)) == 0
    assert cv[1] < 1e-6


# ---- horizons with discrete events (ParNMPCDiscretizer, aux / impulse / lift stages of the backward-Euler formulation) ----
def make_hybrid(N=20, T=1.0, t_lift=0.52, t_touch=0.83):
    """All feet -> {LH, RF} at t_lift -> all feet again at t_touch: one lift and one impulse event off the grid."""
    import ctypes as C
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    p = OracleParNMPC(m, cost, cons, T, N, max_num_impulse=3)
    pts = anymal_contact_points(m).copy()
    p.set_contact_status([1, 1, 1, 1], pts)
    p.push_back_contact_status([0, 1, 1, 0], pts, t_lift)
    p.push_back_contact_status([1, 1, 1, 1], pts, t_touch)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    p.set_solution("q", q)
    p.set_solution("v", v)
    p.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    p.init(0.0)
    return m, p, q, v


def test_hybrid_chain():
    # parnmpc_discretizer.hxx:246-373: the event stages sit in front of the grid stage that follows the event; the stage
    # after the event keeps the rest of the interval
    m, p, q, v = make_hybrid()
    ch = p.chain(0.0)
    dt = 1.0 / 20
    assert "".join(c["kind"][0] for c in ch) == "s" * 10 + "l" + "s" * 6 + "ai" + "s" * 3 + "t"
    lift, aux, imp = ch[10], ch[17], ch[18]
    assert abs(lift["t"] - 0.52) < 1e-12 and abs(lift["dt"] - 0.02) < 1e-12 and abs(ch[11]["dt"] - 0.03) < 1e-12
    assert abs(aux["t"] - 0.83) < 1e-12 and abs(aux["dt"] - 0.03) < 1e-12 and imp["dt"] == 0.0 and abs(ch[19]["dt"] - 0.02) < 1e-12
    assert lift["dimf"] == 12 and ch[11]["dimf"] == 6 and aux["dimf"] == 6 and imp["dimf"] == 6 and ch[19]["dimf"] == 12
    assert [c["level"] for c in (lift, aux, imp)] == [0, 0, -1] and ch[0]["level"] == 1 and ch[-1]["level"] == 20
    assert abs(sum(c["dt"] for c in ch) - 1.0) < 1e-12


def test_hybrid_direction_is_a_newton_direction():
    """With a fixed step size alpha the KKT error of a Newton-type iteration falls by (1 - alpha) per iteration once the
    lagged coupling terms (aux_mat) have caught up: the aux stage with its switching constraint, the impulse stage with its
    own KKT matrix, the lift stage and all the chain neighbour relations have to be consistent for that.  (Full steps from
    this cold start diverge -- ParNMPC has no globalisation; the reference's driver runs 200 iterations.)"""
    m, p, q, v = make_hybrid()
    lib = p.lib
    alpha = 0.1
    e = [p.kkt_error(0.0, q, v)]
    from helpers import P, arr
    for it in range(60):
        for ph in range(5):
            assert lib.oracle_parnmpc_phase(p.h, ph, 0.0, P(arr(q)), P(arr(v))) == 0
        a, b = p.step_sizes()
        lib.oracle_parnmpc_set_step_sizes(p.h, min(a, alpha), min(b, alpha))
        assert lib.oracle_parnmpc_phase(p.h, 5, 0.0, P(arr(q)), P(arr(v))) == 0
        e.append(p.kkt_error(0.0, q, v))
    rate = [e[k + 1] / e[k] for k in range(30, 60)]
    assert max(abs(r - (1 - alpha)) for r in rate) < 5e-3, rate
    assert e[-1] < 0.01 * e[0]


def test_hybrid_parnmpc_converges_to_a_kkt_point():
    """Full (undamped) ParNMPC iterations on chains with a lift, an aux and an impulse stage converge to a KKT point (1e-12): every
    event-stage linearisation, both event-stage KKT inverses, the correction sweeps, expansions and the integration along the
    chain are then mutually consistent.  Three chains: a touch-down onto all four feet, a foot swap (lift-off and touch-down in
    one impulse event) and a chain that STARTS with a lift stage.  (Longer chains -- five and more events -- also converge in
    this restatement, but through transients with KKT errors of 1e4..1e7 in which a perturbation of 1e-13 of the contact points
    decides whether a stage matrix stays positive definite: ParNMPC has no globalisation.  With footholds that move 7.5 cm per
    step the iteration leaves its region of contraction from a standing cold start altogether.)"""
    import ctypes as C
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    for events in ([([0, 1, 1, 0], 0.52), ([1, 1, 1, 1], 0.83)], [([0, 1, 1, 0], 0.52), ([1, 0, 0, 1], 0.83)],
                   [([0, 1, 1, 0], 0.02), ([1, 1, 1, 1], 0.43)]):
        p = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
        pts = anymal_contact_points(m).copy()
        p.set_contact_status([1, 1, 1, 1], pts)
        for status, t_ev in events:
            p.push_back_contact_status(status, pts, t_ev)
        q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
        p.set_solution("q", q)
        p.set_solution("v", v)
        p.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        p.init(0.0)
        e = [p.kkt_error(0.0, q, v)]
        for it in range(40):
            assert p.update(0.0, q, v) == 0
            e.append(p.kkt_error(0.0, q, v))
        assert np.isfinite(e).all() and e[-1] < 1e-10 and max(e) < 1e3, (events, e[::5])
        ch = p.chain(0.0)
        assert sum(c["kind"] == "impulse" for c in ch) == 1 and sum(c["kind"] == "lift" for c in ch) == 1
        # the multiplier of the switching constraint is in play on the aux stage
        xi = p.get_chain("xi", len(ch))
        assert all(np.abs(xi[k]).max() > 0 for k, c in enumerate(ch) if c["kind"] == "aux")
