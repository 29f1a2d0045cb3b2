"""GPU parity of FrictionCone / ImpulseFrictionCone (src/constraints/friction_cone.cpp, impulse_friction_cone.cpp:
two rows per contact, -fz <= 0 and fx^2 + fy^2 - mu^2 fz^2 <= 0) against the oracle, on the three solvers that carry a cone:
OCPSolver on the benchmark's uniform horizon (examples/anymal/ocp_benchmark.cpp:60-118), OCPSolver on a trotting chain with
impulse stages, ParNMPCSolver (examples/anymal/parnmpc_benchmark.cpp).  Bar: 1e-10 on the first Newton direction."""
import numpy as np
import pytest

from idocp_amd import capi
from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC,
                     anymal_contact_points, anymal_model, anymal_problem, rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def start(solvers, m, pts=None, seq=None):
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        if seq is None:
            s.set_contact_status([1, 1, 1, 1], pts)
        else:
            trotting_sequence(s, m, seq)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    return q, v


def test_ocp_benchmark_horizon_direction_iterate_and_ipm_state():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False, cone="nonlinear")
    o, g = OracleOCP(m, cost, cons, 0.5, 20), HipOCP(m, cost, cons, 0.5, 20)
    q, v = start((o, g), m, pts=anymal_contact_points(m))
    o.init_constraints(0.0); g.init_constraints(0.0)
    assert g.lib.idocp_ocp_dimc(g.h) == o.lib.oracle_ocp_dimc(o.h) == 6 * 12 + 2 * 4
    for a, b in zip(g.constraint_data(), o.constraint_data()):
        assert rel_err(a, b) < TOL                                     # setSlackAndDual (friction_cone.cpp:87-97)
    rng = np.random.default_rng(3)
    q[7:] += 0.02 * rng.uniform(-1, 1, 12)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    errs = []
    for it in range(25):
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
        errs.append((o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]))
    assert errs[-1][1] < 1e-6 * e_g and abs(np.log10(errs[-1][1] / errs[-1][0])) < 1.0, errs[-3:]


def test_trotting_chain_with_impulse_cone():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True, cone="nonlinear")
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    q, v = start((o, g), m, seq=nimp)
    o.init_constraints(0.0); g.init_constraints(0.0)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-9, f
    for f in ("q", "v", "a", "f", "u", "lmd", "gmm", "beta", "mu"):
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-9, f
    for _ in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    assert g.kkt_error(0.0, q, v)[0] < 1e-6 * e_g


def test_parnmpc_benchmark_horizon():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False, cone="nonlinear")
    o, g = OracleParNMPC(m, cost, cons, 0.5, 20), HipParNMPC(m, cost, cons, 0.5, 20)
    q, v = start((o, g), m, pts=anymal_contact_points(m))
    o.init(0.0); g.init(0.0)
    q[7:] += 0.05
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    for f in OCP_SOL_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    for _ in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    assert abs(g.kkt_error(0.0, q, v)[0] - o.kkt_error(0.0, q, v)) < 1e-6 and g.kkt_error(0.0, q, v)[0] < 1e-3 * e_g[0]


def test_both_cones_together_are_rejected():
    import ctypes as C
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False, cone="nonlinear")
    cons.linearized_friction_cone = 1
    h = C.c_void_p()
    rc = capi.lib().idocp_ocp_create(C.byref(m), C.byref(cost), C.byref(cons), 0.5, 20, 1, 0, C.byref(h))
    assert rc == -4 and b"FrictionCone" in capi.lib().idocp_last_error()
