"""OCPSolver / ParNMPCSolver on a FIXED-BASE robot without contacts -- the reference's examples/iiwa14/ocp_benchmark.cpp and
parnmpc_benchmark.cpp -- restated by the oracle (oracle/ocp.cpp with `kP = 0` passive rows, the `has_floating_base_ == false` branches of
state_equation.hxx:20-200, contact_dynamics.hxx:65-190, backward_riccati_recursion_factorizer.hxx:48-150, split_kkt_matrix_inverter.hxx:100-190).

With no contact rows the contact-dynamics formulation (control u, the acceleration eliminated through M^-1) and the unconstrained one of
UnOCPSolver / UnParNMPCSolver (control a, the torque eliminated through u = ID(q, v, a)) condense ONE Newton system in two orders: same
direction, same step sizes, same KKT error.  The two restatements share the rigid-body layer only (`ocp.cpp` against `unocp.cpp`), so their
agreement to rounding checks each against the other -- and it is what lets the product bind `idocp::OCPSolver` on such a robot to the
fixed-base kernels (include/idocp/ocp/ocp_solver.hpp; tests/test_fixed_base_ocp_gpu.py holds the GPU to THIS restatement)."""
import numpy as np

from helpers import OracleOCP, OracleParNMPC, OracleUnOCP, OracleUnParNMPC, iiwa14_model, unocp_problem

FIELDS = ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta")


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def start(model):
    return np.full(model.nq, 2.0), np.zeros(model.nv)      # examples/iiwa14/ocp_benchmark.cpp:44-45


def test_ocp_solver_on_a_fixed_base_robot_is_the_unconstrained_solver():
    model = iiwa14_model()
    cost, cons = unocp_problem(model)                       # the cost and limits of ocp_benchmark.cpp:21-36 (= unocp_benchmark.cpp)
    T, N = 1.0, 20
    q, v = start(model)
    o, u = OracleOCP(model, cost, cons, T, N), OracleUnOCP(model, cost, cons, T, N)
    for s in (o, u):
        s.set_solution("q", q)
        s.set_solution("v", v)
    o.init_constraints(0.0)                                 # (UnOCPSolver::setSolution initialises them itself, unocp_solver.cpp:180)
    e_o, e_u = o.kkt_error(0.0, q, v), u.kkt_error(0.0, q, v)
    assert abs(e_o - e_u) <= 1e-12 * e_u
    for it, tol in enumerate((1e-11, 1e-10, 1e-10)):      # (dbeta, ~1e-6 with a zero torque weight, is a difference of terms of order 0.1 behind M^-1)
        o.update(0.0, q, v)
        u.update(0.0, q, v)
        for name in FIELDS:
            a, b = o.get(name), u.direction(name)
            assert a.shape == b.shape, name
            assert rel(a, b) <= tol, (it, name, rel(a, b))
        assert np.allclose(o.step_sizes(), u.step_sizes(), rtol=1e-9, atol=0)
        for name in ("q", "v", "a", "u", "lmd", "gmm", "beta"):
            assert rel(o.get(name), u.solution(name)) <= 10 * tol, (it, name)
        e_o, e_u = o.kkt_error(0.0, q, v), u.kkt_error(0.0, q, v)
        assert abs(e_o - e_u) <= 1e-9 * e_u, (it, e_o, e_u)


def test_the_policy_of_the_fixed_base_ocp_solver_is_the_acceleration_policy_mapped_through_the_inverse_dynamics():
    """du = K dx + k of the contact-dynamics formulation (lqr_state_feedback_policy.hpp) against UnOCPSolver's da = Ka dx + ka:
    K = [dID/dq + M Ka_q, dID/dv + M Ka_v] (unconstrained_dynamics.hxx:84-92, du = ID + dID/dq dq + dID/dv dv + M da), checked on
    the directions themselves: du - K dx - k = 0 stage by stage for the oracle's OCP gains."""
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    T, N = 1.0, 20
    q, v = start(model)
    o = OracleOCP(model, cost, cons, T, N)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init_constraints(0.0)
    o.update(0.0, q, v)
    _, _, K, k = o.riccati()                                # K[N][nu][2nv]
    dq, dv, du = o.get("dq"), o.get("dv"), o.get("du")
    for i in range(N):
        dx = np.concatenate([dq[i], dv[i]])
        assert np.abs(du[i] - K[i] @ dx - k[i]).max() <= 1e-9 * max(1.0, np.abs(du[i]).max()), i


def test_parnmpc_solver_on_a_fixed_base_robot_is_the_unconstrained_parnmpc_solver():
    model = iiwa14_model()
    cost, cons = unocp_problem(model)                       # parnmpc_benchmark.cpp:21-36
    T, N = 1.0, 20
    q, v = start(model)
    o, u = OracleParNMPC(model, cost, cons, T, N), OracleUnParNMPC(model, cost, cons, T, N)
    for s in (o, u):
        s.set_solution("q", q)
        s.set_solution("v", v)
    o.init(0.0)                                             # initBackwardCorrection(t), parnmpc_benchmark.cpp:51
    u.init(0.0)
    e_o, e_u = o.kkt_error(0.0, q, v), u.kkt_error(0.0, q, v)
    assert abs(e_o - e_u) <= 1e-12 * e_u
    for it, tol in enumerate((1e-11, 1e-9, 1e-9)):
        o.update(0.0, q, v)
        u.update(0.0, q, v)
        for name in FIELDS + ("q", "v", "a", "u", "lmd", "gmm", "beta"):
            a, b = o.get(name), u.get(name)
            assert a.shape == b.shape, name
            assert rel(a, b) <= tol, (it, name, rel(a, b))
        assert np.allclose(o.step_sizes(), u.step_sizes(), rtol=1e-8, atol=0)
