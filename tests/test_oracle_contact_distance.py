"""Oracle: ContactDistance (src/constraints/contact_distance.cpp) as IPM component 10 of the contact-capable solvers: the frames of the
contacts that are not active stay above z = 0 (rows of the active contacts idle with dslack = ddual = 1)."""
import numpy as np

from helpers import ANYMAL_Q_STANDING, OracleOCP, anymal_contact_points, anymal_model, anymal_problem, trotting_sequence


def make(N=31, T=1.55, nimp=2):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    cons.contact_distance = 1
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    trotting_sequence(o, m, nimp)
    o.set_solution("q", q); o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    return m, o, q, v


def test_slack_is_the_height_of_the_contact_frames():
    m, o, q, v = make()
    sl, du = o.constraint_data()
    assert sl.shape[1] == 6 * 12 + 5 * 4 + 4
    # setSlackAndDual (contact_distance.cpp:58-65): z of the four feet of the standing robot -- on the ground, i.e. pushed up to the
    # barrier parameter by setSlackAndDualPositive; the position level starts at stage 2
    z = anymal_contact_points(m)[:, 2]
    assert np.abs(z).max() < 1e-3
    assert np.all(sl[:2, 92:] == 0) and np.all(sl[2:20, 92:] > 0) and np.all(sl[2:20, 92:] <= 2e-4 + np.abs(z).max())


def test_swing_feet_stay_above_the_ground_and_the_sqp_converges():
    m, o, q, v = make()
    e0 = o.kkt_error(0.0, q, v)
    for _ in range(40):
        assert o.update(0.0, q, v) == 0
    e1 = o.kkt_error(0.0, q, v)
    assert e1 < 1e-12 * e0, (e0, e1)                       # 207 -> 5e-14, full steps from iteration 20 on
    assert o.infeasible_stage() == -1
    sl, du = o.constraint_data()
    assert sl[13, 92] > 0.02 and sl[13, 95] > 0.02        # LF and RH in the air on grid stage 13 (slack = height at convergence)
    assert np.allclose(sl[13, [92, 95]] * du[13, [92, 95]], 1e-4, rtol=1e-6)      # complementarity at the barrier value
