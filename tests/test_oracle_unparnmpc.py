"""CPU checks of the UnParNMPC half of the oracle (src/unocp/unparnmpc_solver.cpp, unbackward_correction.cpp):
the block formula of the per-stage KKT inverse against a dense inverse (structure of the recovered KKT matrix), the coarse
update as the stage-wise Newton step, the constraint gating, and convergence under the protocol of
examples/iiwa14/unparnmpc_benchmark.cpp."""
import numpy as np

from helpers import OracleUnParNMPC, iiwa14_model, unocp_problem


def make(N=20, T=1.0):
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnParNMPC(m, cost, cons, T, N)
    q, v = np.full(m.nv, 2.0), np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init(0.0)
    return m, o, q, v


def test_kkt_inverse_blocks_equal_the_dense_inverse():
    # split_unkkt_matrix_inverter.hxx:37-80: K^-1 of [[0 F] [F^T Q]] with F = [0 -I dt I; dt I 0 -I]
    m, o, q, v = make()
    nv, dt = m.nv, 1.0 / 20
    assert o.stage(0, 0.0, q, v) == 0
    Kinv, aux = o.matrices()
    Z, I = np.zeros((nv, nv)), np.eye(nv)
    F = np.block([[Z, -I, dt * I], [dt * I, Z, -I]])
    for i in (0, 7, 19):
        Ki = Kinv[i]
        assert np.abs(Ki - Ki.T).max() < 1e-9 * np.abs(Ki).max()
        # recover the KKT matrix from its inverse: the constraint blocks must be exactly F, the (1,1) block zero
        Kmat = np.linalg.inv(Ki)
        scale = np.abs(Kmat).max()
        assert np.abs(Kmat[:2 * nv, :2 * nv]).max() < 1e-8 * scale
        assert np.abs(Kmat[:2 * nv, 2 * nv:] - F).max() < 1e-8 * scale
        Q = Kmat[2 * nv:, 2 * nv:]
        assert np.linalg.eigvalsh(0.5 * (Q + Q.T)).min() > 0
    # initAuxMat: the terminal cost Hessian on every stage
    assert np.allclose(aux[3], np.diag(np.r_[np.full(nv, 10.0), np.full(nv, 0.1)]))


def test_coarse_update_is_the_stagewise_newton_step():
    m, o, q, v = make()
    assert o.stage(0, 0.0, q, v) == 0
    for f in ("lmd", "gmm", "a", "q", "v"):
        assert np.allclose(o.get("new_" + f), o.get(f) - o.get("d" + f), rtol=0, atol=1e-12)


def test_convergence_on_the_reference_example():
    # the four sweeps solve the banded KKT system approximately (the coupling through aux_mat is the previous iterate's):
    # iterating drives the KKT error to its floor and the trajectory onto the backward-Euler dynamics
    m, o, q, v = make()
    errs = [o.kkt_error(0.0, q, v)]
    for _ in range(120):      # num_iteration = 100 in examples/iiwa14/unparnmpc_benchmark.cpp; the floor (6e-8) is reached after ~90
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    assert errs[-1] < 1e-6 and errs[-1] < 1e-8 * errs[0]
    assert o.infeasible_stage() == -1
    # backward-Euler feasibility of the converged trajectory
    dt = 1.0 / 20
    Q, V, A = o.get("q"), o.get("v"), o.get("a")
    qp, vp = np.vstack([q, Q[:-1]]), np.vstack([v, V[:-1]])
    assert np.abs(qp - Q + dt * V).max() < 1e-8
    assert np.abs(vp - V + dt * A).max() < 1e-8


def test_gating_and_step_sizes():
    m, o, q, v = make()
    sl, du = o.constraint_data()
    nv = m.nv
    assert np.all(sl[0, :2 * nv] == 0) and np.all(sl[1, :2 * nv] > 0)      # stage 0 is created with time step 1: no position rows
    assert np.all(sl[0, 2 * nv:] > 0)
    assert o.update(0.0, q, v) == 0
    ap, ad = o.step_sizes()
    assert 0 < ap <= 1 and 0 < ad <= 1


def test_filter_line_search_step_rule():
    # UnLineSearch::computeStepSize (unline_search.hpp:62-92): the accepted step is the fraction-to-boundary step times a
    # power of 0.75, or the floor 0.05 -- also when the fraction-to-boundary step itself is below the floor, which can push
    # a slack through zero (the barrier cost is NaN from then on and the filter accepts everything: reference behaviour)
    from helpers import OracleUnOCP
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
    for make_solver in (lambda: OracleUnOCP(m, cost, cons, 1.0, 20), lambda: OracleUnParNMPC(m, cost, cons, 1.0, 20)):
        o = make_solver()
        o.set_solution("q", q)
        o.set_solution("v", v)
        if hasattr(o, "init"):
            o.init(0.0)
        c0 = o.cost_and_violation(0.0, q, v) if isinstance(o, OracleUnParNMPC) else o.cost_and_violation(0.0)
        assert np.isfinite(c0).all() and c0[0] > 0 and c0[1] > 0
        for it in range(6):
            assert o.update(0.0, q, v, line_search=True) == 0
            assert 0.05 <= o.step_sizes()[0] <= 1.0
        o.clear_line_search_filter()
        assert o.update(0.0, q, v, line_search=True) == 0
