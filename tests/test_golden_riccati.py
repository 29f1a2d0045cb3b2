"""The oracle's backward Riccati recursion (oracle/ocp.cpp OCPSolver::backwardRiccatiRecursion, restating
backward_riccati_recursion_factorizer.hxx:44-161 / split_riccati_factorizer.hxx:24-52 / riccati_recursion_solver.cpp:48-107) against
an INDEPENDENTLY generated golden fixture: tests/golden/riccati_lqr.json holds a random linear-quadratic problem in the reference's
block structure and, for every stage, P, s, K, k read off the solution of the DENSE KKT system of the tail problem
(tests/golden/gen_golden_riccati.py: numpy, no recursion, no formula shared with the oracle -- the counterpart of the reference's
test/ocp/riccati_recursion_solver_test.cpp).  The stage data are injected into an oracle solver of the same dimensions (ANYmal,
nv = 18, nu = 12) and only the sweep runs.  CPU only."""
import ctypes as C
import json
import os

import numpy as np

from helpers import ANYMAL_Q_STANDING, GOLDEN, OracleOCP, P, anymal_contact_points, anymal_model, anymal_problem, arr, dp


def test_backward_riccati_recursion_equals_the_dense_kkt_solution():
    g = json.load(open(os.path.join(GOLDEN, "riccati_lqr.json")))
    nv, nu, N = g["nv"], g["nu"], g["N"]
    m = anymal_model()
    assert m.nv == nv and m.nu == nu
    cost, cons = anymal_problem(m, trotting_ref=False)
    dt = g["stages"][0]["dt"]
    o = OracleOCP(m, cost, cons, dt * N, N)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(nv)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    assert o.update(0.0, q, v) == 0                      # discretises the horizon and sizes every stage block
    lib = o.lib
    lib.oracle_ocp_inject_lqr_stage.argtypes = [C.c_void_p, C.c_int, C.c_int] + [dp] * 11
    lib.oracle_ocp_backward_riccati_only.argtypes = [C.c_void_p]
    cm = lambda a: arr(np.asarray(a, dtype=np.float64).T)      # column-major
    for i, st in enumerate(g["stages"]):
        assert lib.oracle_ocp_inject_lqr_stage(o.h, i, 0, P(cm(st["Qxx"])), P(cm(st["Qxu"])), P(cm(st["Quu"])), P(cm(st["Fqq6"])), P(cm(st["Fqv6"])),
                                               P(cm(st["Fvq"])), P(cm(st["Fvv"])), P(cm(st["Fvu"])), P(arr(st["lx"])), P(arr(st["lu"])),
                                               P(arr(st["Fx"]))) == 0
    t = g["terminal"]
    z = P(np.zeros(4))
    assert lib.oracle_ocp_inject_lqr_stage(o.h, N, 1, P(cm(t["Qxx"])), z, z, z, z, z, z, z, P(arr(t["lx"])), z, z) == 0
    assert lib.oracle_ocp_backward_riccati_only(o.h) == 0
    Pm, s, K, k = o.riccati()
    for i in range(N + 1):
        r = g["riccati"][i]
        Pg, sg = np.asarray(r["P"]), np.asarray(r["s"])
        # the dense solve itself is good to cond(KKT) eps ~ 1e-12 of the entries
        assert np.abs(Pm[i] - Pg).max() <= 1e-10 * max(1.0, np.abs(Pg).max()), (i, np.abs(Pm[i] - Pg).max())
        assert np.abs(s[i] - sg).max() <= 1e-10 * max(1.0, np.abs(sg).max()), (i, np.abs(s[i] - sg).max())
        if i < N:
            Kg, kg = np.asarray(r["K"]), np.asarray(r["k"])
            assert np.abs(K[i] - Kg).max() <= 1e-10 * max(1.0, np.abs(Kg).max()), (i, np.abs(K[i] - Kg).max())
            assert np.abs(k[i] - kg).max() <= 1e-10 * max(1.0, np.abs(kg).max()), (i, np.abs(k[i] - kg).max())
