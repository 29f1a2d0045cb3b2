"""The oracle's contact-path OCP layer (ANYmal, uniform 4-contact horizon) against
the dense-formula identities of the reference's unit tests (CPU only)."""
import numpy as np

from helpers import ANYMAL_Q_STANDING, OracleOCP, anymal_contact_points, anymal_model, anymal_problem


def make(N=20, T=1.0, trotting_ref=False, perturb=0.0):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=trotting_ref)
    o = OracleOCP(m, cost, cons, T, N)
    pts = anymal_contact_points(m)
    o.set_contact_status([1, 1, 1, 1], pts)
    q = ANYMAL_Q_STANDING.copy()
    v = np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    if perturb:
        rng = np.random.default_rng(3)
        q[7:] += perturb * rng.uniform(-1, 1, 12)
        q[0:2] += perturb * rng.uniform(-1, 1, 2)
    return m, o, q, v, pts


def test_contact_points_are_the_feet_on_the_ground():
    m = anymal_model()
    pts = anymal_contact_points(m)
    assert np.abs(pts[:, 2]).max() < 0.02              # standing pose: feet at ground height
    assert pts[0, 0] > 0 and pts[1, 0] < 0 and pts[0, 1] > 0 and pts[2, 1] < 0      # LF, LH, RF, RH


def test_convergence_standing():
    # ocpbenchmarker::Convergence protocol on the uniform 4-contact problem (examples/anymal/ocp_benchmark.cpp:99-118)
    m, o, q, v, _ = make(N=20, T=0.5, perturb=0.02)
    e0 = o.kkt_error(0.0, q, v)
    errs = [e0]
    for _ in range(30):
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    assert np.isfinite(errs).all()
    assert errs[-1] < 1e-4 * e0, errs[-5:]


def test_riccati_backward_matches_dense_lqr_formulas():
    # test/ocp/backward_riccati_recursion_factorizer_test.cpp:113-165, test/ocp/split_riccati_factorizer_test.cpp
    m, o, q, v, _ = make(N=8, T=0.4, trotting_ref=True, perturb=0.03)
    o.update(0.3, q, v)
    o.update(0.3, q, v)
    assert o.stage(0, 0.3, q, v) == 0
    lqr = [o.lqr_stage(i) for i in range(o.N)]        # before the sweep modifies the blocks
    assert o.stage(1, 0.3, q, v) == 0
    P, s, K, k = o.riccati()
    for i in range(o.N - 1, -1, -1):
        Qxx, Qxu, Quu, A, B, lx, lu, Fx = lqr[i]
        Qxx = np.triu(Qxx) + np.triu(Qxx, 1).T if False else Qxx
        Pn, sn = P[i + 1], s[i + 1]
        F = Qxx + A.T @ Pn @ A
        H = Qxu + A.T @ Pn @ B
        G = Quu + B.T @ Pn @ B
        Kd = -np.linalg.solve(G, H.T)
        kd = -np.linalg.solve(G, lu + B.T @ (Pn @ Fx - sn))
        np.testing.assert_allclose(K[i], Kd, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(k[i], kd, rtol=1e-8, atol=1e-8)
        Pd = F - Kd.T @ G @ Kd
        np.testing.assert_allclose(P[i], 0.5 * (Pd + Pd.T), rtol=1e-8, atol=1e-7)
        sd = A.T @ (sn - Pn @ Fx) - lx - H @ kd
        np.testing.assert_allclose(s[i], sd, rtol=1e-8, atol=1e-7)


def test_forward_recursion_costate_and_feasible_full_step():
    m, o, q, v, _ = make(N=8, T=0.4, perturb=0.03)
    o.update(0.0, q, v)
    assert o.stage(0, 0.0, q, v) == 0
    lqr = [o.lqr_stage(i) for i in range(o.N)]
    assert o.stage(1, 0.0, q, v) == 0
    assert o.stage(2, 0.0, q, v) == 0
    P, s, K, k = o.riccati()
    dq, dv, du = o.get("dq"), o.get("dv"), o.get("du")
    dl, dg = o.get("dlmd"), o.get("dgmm")
    for i in range(o.N):
        _, _, _, A, B, _, _, Fx = lqr[i]
        dx = np.concatenate([dq[i], dv[i]])
        np.testing.assert_allclose(du[i], K[i] @ dx + k[i], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(np.concatenate([dq[i + 1], dv[i + 1]]), A @ dx + B @ du[i] + Fx, rtol=1e-10, atol=1e-10)
    for i in range(o.N + 1):
        dx = np.concatenate([dq[i], dv[i]])
        np.testing.assert_allclose(np.concatenate([dl[i], dg[i]]), P[i] @ dx - s[i], rtol=1e-9, atol=1e-8)
    a, b = o.step_sizes()
    assert 0 < a <= 1 and 0 < b <= 1


def test_openmp_stage_loops_give_the_single_thread_result():
    """The oracle's stage loops run under `#pragma omp parallel for num_threads(nthreads)` where the reference's do
    (ocp_linearizer.cpp:74-83, unocp_solver.cpp:78-94): per-stage work is independent, so 1 and 4 threads must agree bit for bit."""
    import numpy as np
    from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OracleOCP, anymal_contact_points, anymal_model, anymal_problem, oracle)
    lib = oracle()
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    outs = []
    for nt in (1, 4):
        o = OracleOCP(m, cost, cons, 1.0, 20)
        lib.oracle_ocp_set_num_threads(o.h, nt)
        q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
        o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
        o.set_solution("q", q)
        o.set_solution("v", v)
        o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        o.init_constraints(0.0)
        q[7:] += 0.02
        for _ in range(3):
            assert o.update(0.0, q, v) == 0
        outs.append([o.get(f) for f in OCP_DIR_FIELDS] + [np.array(o.step_sizes())])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
