"""Oracle: JointAccelerationLowerLimit / UpperLimit (src/constraints/joint_acceleration_{lower,upper}_limit.cpp) as IPM components 8 / 9
of the contact-capable solvers."""
import numpy as np

from helpers import ANYMAL_Q_STANDING, OracleOCP, anymal_contact_points, anymal_model, anymal_problem


def make(a_max, lower=1, upper=1):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    cons.joint_acceleration_lower_limit = lower
    cons.joint_acceleration_upper_limit = upper
    for j in range(12):
        cons.a_min[j] = -a_max
        cons.a_max[j] = a_max
    o = OracleOCP(m, cost, cons, 0.5, 20)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    o.set_solution("q", q); o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    rng = np.random.default_rng(5)
    q[7:] += 0.005 * rng.uniform(-1, 1, 12)       # the free problem answers with |a| up to 12 rad/s^2
    return m, o, q, v


def test_slack_initialisation_and_dimension():
    m, o, q, v = make(2.5)
    sl, du = o.constraint_data()
    assert sl.shape[1] == 6 * 12 + 5 * 4 + 2 * 12
    # setSlackAndDual (joint_acceleration_upper_limit.cpp:50-54): slack = amax - a = 2.5 at a = 0, dual = barrier / slack
    assert np.allclose(sl[:, 92:], 2.5) and np.allclose(du[:, 92:], 1e-4 / 2.5)


def test_the_bound_binds_and_the_sqp_converges():
    m, o, q, v = make(1e3)
    for _ in range(30):
        assert o.update(0.0, q, v) == 0
    a_free = np.abs(o.get("a")[:, 6:]).max()
    assert a_free > 10.0                                  # the free problem accelerates harder than the bound used below
    m, o, q, v = make(6.0)
    e0 = o.kkt_error(0.0, q, v)
    for _ in range(40):
        assert o.update(0.0, q, v) == 0
    e1 = o.kkt_error(0.0, q, v)
    a = o.get("a")[:, 6:]
    assert e1 < 1e-6 * e0 and np.abs(a).max() <= 6.0 + 1e-9 and np.abs(a).max() > 5.99
    sl, du = o.constraint_data()
    # complementarity at the barrier value on every row
    assert np.allclose(sl[:, 92:] * du[:, 92:], 1e-4, rtol=1e-3)
