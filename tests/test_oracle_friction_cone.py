"""FrictionCone / ImpulseFrictionCone in the CPU restatement: the rows of the component against the
reference's formulas, and the SQP with the cone of examples/anymal/ocp_benchmark.cpp (CPU only)."""
import ctypes as C

import numpy as np

from helpers import ANYMAL_Q_STANDING, OracleOCP, P, anymal_contact_points, anymal_model, anymal_problem, oracle


def cone_eval(kind, mu, f):
    lib = oracle()
    lib.oracle_cone_eval.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oracle_cone_eval.restype = C.c_int
    res, J = np.zeros(5), np.zeros(15)
    f = np.ascontiguousarray(f, dtype=np.float64)
    nr = lib.oracle_cone_eval(kind, mu, P(f), P(res), P(J))
    return nr, res[:nr], J.reshape(5, 3)[:nr]


def test_rows_of_the_nonlinear_cone_follow_the_reference_formulas():
    # friction_cone.hpp:70-80: frictionConeResidual = fx^2 + fy^2 - mu^2 fz^2, normalForceResidual = -fz;
    # friction_cone.cpp:100-118: data.r = (2 fx, 2 fy, -2 mu^2 fz) is what multiplies the dual and builds Qff
    rng = np.random.default_rng(5)
    for _ in range(20):
        mu = rng.uniform(0.3, 1.0)
        f = rng.uniform(-50, 50, 3)
        nr, res, J = cone_eval(1, mu, f)
        assert nr == 2
        assert abs(res[0] - (-f[2])) < 1e-15 * max(1.0, abs(f[2]))
        assert abs(res[1] - (f[0] ** 2 + f[1] ** 2 - mu ** 2 * f[2] ** 2)) < 1e-12 * max(1.0, abs(res[1]))
        assert np.allclose(J[0], [0, 0, -1], atol=0)
        assert np.allclose(J[1], [2 * f[0], 2 * f[1], -2 * mu ** 2 * f[2]], rtol=1e-15)
        # the gradient is the derivative of the residual (central differences)
        h = 1e-6
        for x in range(3):
            e = np.zeros(3); e[x] = h
            _, rp, _ = cone_eval(1, mu, f + e)
            _, rm, _ = cone_eval(1, mu, f - e)
            assert np.allclose((rp - rm) / (2 * h), J[:, x], rtol=1e-6, atol=1e-6)


def test_rows_of_the_linearized_cone_are_unchanged():
    # linearized_friction_cone.cpp:25-29, linearized_friction_cone.hpp:72-84
    mu, f = 0.7, np.array([3.0, -2.0, 40.0])
    nr, res, J = cone_eval(0, mu, f)
    m2 = mu / np.sqrt(2.0)
    Jc = np.array([[0, 0, -1], [1, 0, -m2], [-1, 0, -m2], [0, 1, -m2], [0, -1, -m2]])
    assert nr == 5 and np.allclose(J, Jc, atol=1e-16) and np.allclose(res, Jc @ f, atol=1e-13)


def make(N, T, cone):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False, cone=cone)
    o = OracleOCP(m, cost, cons, T, N)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    q = ANYMAL_Q_STANDING.copy()
    o.set_solution("q", q)
    o.set_solution("v", np.zeros(m.nv))
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    rng = np.random.default_rng(3)
    q[7:] += 0.02 * rng.uniform(-1, 1, 12)
    return m, o, q, np.zeros(m.nv)


def test_sqp_with_the_cone_of_the_anymal_benchmark_converges():
    # ocpbenchmarker::Convergence protocol of examples/anymal/ocp_benchmark.cpp:60-118 (N = 20, T = 0.5, FrictionCone(mu = 0.7))
    m, o, q, v = make(20, 0.5, "nonlinear")
    sl, du = o.constraint_data()
    assert sl.shape[1] == 6 * 12 + 2 * 4                  # dimc: two rows per contact (friction_cone.cpp:12)
    fz = 0.25 * (-m.total_mass * m.gravity[2])
    # setSlackAndDual (friction_cone.cpp:87-97): slack = (fz, mu^2 fz^2 - fx^2 - fy^2) at the initial guess f = (0, 0, fz)
    assert np.allclose(sl[5, 72:], np.tile([fz, 0.49 * fz * fz], 4), rtol=1e-12)
    e0 = o.kkt_error(0.0, q, v)
    errs = [e0]
    for _ in range(40):
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    assert np.isfinite(errs).all() and errs[-1] < 1e-4 * e0, errs[-5:]
    # the linearized cone is an inner approximation: the same problem converges to a different, nearby point
    m2, o2, _, _ = make(20, 0.5, "linearized")
    for _ in range(40):
        assert o2.update(0.0, q, v) == 0
    assert np.abs(o.get("f") - o2.get("f")).max() < 5.0
