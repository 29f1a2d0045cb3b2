"""The oracle's OCP layer against the dense-formula identities that the
reference's own unit tests assert (CPU only).  The reference holds no numeric
vectors for this layer, so these identities plus the rigid-body
golden vectors are what pins the restatement."""
import numpy as np

from helpers import OracleUnOCP, iiwa14_model, unocp_problem


def make(N=20, T=1.0):
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnOCP(m, cost, cons, T, N)
    q = np.full(m.nv, 2.0)
    v = np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    return m, o, q, v


def test_convergence_config_c1():
    # examples/iiwa14/unocp_benchmark.cpp:40-53: T=1, N=20, q=2, v=0; ocpbenchmarker::Convergence
    m, o, q, v = make()
    e0 = o.kkt_error(0.0, q, v)
    errs = [e0]
    for _ in range(40):
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    assert np.isfinite(errs).all()
    assert errs[-1] < 1e-6 * e0
    assert errs[-1] < 1e-3


def test_riccati_backward_matches_dense_formulas():
    # test/unocp/backward_unriccati_recursion_factorizer_test.cpp:64-108 and
    # test/unocp/split_unriccati_factorizer_test.cpp:65-145
    m, o, q, v = make(N=12, T=0.6)
    o.update(0.0, q, v)          # move away from the trivial initial guess
    o.update(0.0, q, v)
    assert o.stage(0, 0.0, q, v) == 0
    Qraw, res = o.unkkt()        # condensed stage KKT before the backward sweep touches it
    assert o.stage(1, 0.0, q, v) == 0
    P, s, K, k = o.riccati()
    nv, N, dt = m.nv, o.N, 0.6 / 12
    I, Z = np.eye(nv), np.zeros((nv, nv))
    A = np.block([[I, dt * I], [Z, I]])
    B = np.vstack([Z, dt * I])
    for i in range(N - 1, -1, -1):
        Q = Qraw[i]
        Q = np.triu(Q) + np.triu(Q, 1).T        # only the upper blocks are written by the condensation
        Qaa, Qax, Qxx = Q[:nv, :nv], Q[:nv, nv:], Q[nv:, nv:]
        Fx, la, lx = res[i, :2 * nv], res[i, 2 * nv:3 * nv], res[i, 3 * nv:]
        Pn, sn = P[i + 1], s[i + 1]
        F = Qxx + A.T @ Pn @ A
        H = Qax.T + A.T @ Pn @ B
        G = Qaa + B.T @ Pn @ B
        la_f = la + B.T @ Pn @ Fx - B.T @ sn
        Kd = -np.linalg.solve(G, H.T)
        kd = -np.linalg.solve(G, la_f)
        np.testing.assert_allclose(K[i], Kd, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(k[i], kd, rtol=1e-9, atol=1e-9)
        Pd = F - Kd.T @ G @ Kd
        np.testing.assert_allclose(P[i], Pd, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(P[i], P[i].T, atol=1e-12)      # P symmetric
        sd = A.T @ sn - A.T @ Pn @ Fx - lx - H @ kd
        np.testing.assert_allclose(s[i], sd, rtol=1e-9, atol=1e-8)


def test_forward_recursion_and_costate():
    m, o, q, v = make(N=12, T=0.6)
    o.update(0.0, q, v)
    q2 = q + 0.05
    assert o.stage(0, 0.0, q2, v) == 0
    _, res = o.unkkt()
    assert o.stage(1, 0.0, q2, v) == 0
    assert o.stage(2, 0.0, q2, v) == 0
    P, s, K, k = o.riccati()
    nv, N, dt = m.nv, o.N, 0.6 / 12
    dq, dv, da = o.direction("dq"), o.direction("dv"), o.direction("da")
    dl, dg = o.direction("dlmd"), o.direction("dgmm")
    np.testing.assert_allclose(dq[0], q2 - o.solution("q")[0], atol=1e-14)
    for i in range(N):
        dx = np.concatenate([dq[i], dv[i]])
        np.testing.assert_allclose(da[i], K[i] @ dx + k[i], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(dq[i + 1], dq[i] + dt * dv[i] + res[i, :nv], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(dv[i + 1], dv[i] + dt * da[i] + res[i, nv:2 * nv], rtol=1e-12, atol=1e-12)
    for i in range(N + 1):
        dx = np.concatenate([dq[i], dv[i]])
        np.testing.assert_allclose(np.concatenate([dl[i], dg[i]]), P[i] @ dx - s[i], rtol=1e-10, atol=1e-9)


def test_step_sizes_in_unit_interval_and_slack_positive():
    m, o, q, v = make()
    for _ in range(5):
        o.update(0.0, q, v)
        a, b = o.step_sizes()
        assert 0 < a <= 1 and 0 < b <= 1
    sl, du = o.constraint_data()
    # rows of levels that are not valid at stages 0/1 read 0 (constraints_data.hpp:18-42)
    assert (sl[2:] > 0).all() and (du[2:] > 0).all()
    assert (sl[0, :4 * m.nv] == 0).all() and (sl[1, :2 * m.nv] == 0).all() and (sl[1, 2 * m.nv:] > 0).all()
