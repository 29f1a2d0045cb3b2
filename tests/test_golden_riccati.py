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


import pytest      # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [0, 1])
def test_hip_backward_riccati_sweep_equals_the_dense_kkt_solution(sweep, monkeypatch):
    """The same fixture against the KERNEL: the stage blocks go into the device records (idocp_ocp_set_lqr_stage), the backward sweep runs alone
    (idocp_ocp_launch_kernel id 2) in both of its forms -- one wavefront per instance with P in registers (the headline form), eight per instance with
    P in LDS -- and P, s, K, k of every stage are held to the dense KKT solution, with no restatement in between."""
    from helpers import HipOCP, force_forms
    from idocp_amd import capi
    force_forms(monkeypatch, sweep=sweep)
    g = json.load(open(os.path.join(GOLDEN, "riccati_lqr.json")))
    nv, nu, N = g["nv"], g["nu"], g["N"]
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    dt = g["stages"][0]["dt"]
    h = HipOCP(m, cost, cons, dt * N, N, batch=3)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(nv)
    h.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    h.set_solution("q", q)
    h.set_solution("v", v)
    h.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    h.init_constraints(0.0)
    assert h.update(0.0, q, v) == 0                      # discretises the horizon
    assert h.lib.idocp_ocp_riccati_sweep(h.h) == sweep
    cm = lambda a: arr(np.asarray(a, dtype=np.float64).T)      # column-major
    for i, st in enumerate(g["stages"]):
        capi.check(h.lib.idocp_ocp_set_lqr_stage(h.h, i, 0, P(cm(st["Qxx"])), P(cm(st["Qxu"])), P(cm(st["Quu"])), P(cm(st["Fqq6"])), P(cm(st["Fqv6"])),
                                                 P(cm(st["Fvq"])), P(cm(st["Fvv"])), P(cm(st["Fvu"])), P(arr(st["lx"])), P(arr(st["lu"])),
                                                 P(arr(st["Fx"]))), "set_lqr_stage")
    t = g["terminal"]
    capi.check(h.lib.idocp_ocp_set_lqr_stage(h.h, N, 1, P(cm(t["Qxx"])), None, None, None, None, None, None, None, P(arr(t["lx"])), None, None), "terminal")
    dq, dv = C.c_void_p(), C.c_void_p()                  # (the launch entry wants the measured state on the device; the backward sweep does not read it)
    qb, vb = arr(np.tile(q, (3, 1))), arr(np.tile(v, (3, 1)))
    capi.check(h.lib.idocp_device_alloc(C.byref(dq), qb.nbytes), "alloc")
    capi.check(h.lib.idocp_device_alloc(C.byref(dv), vb.nbytes), "alloc")
    capi.check(h.lib.idocp_device_upload(dq, qb.ctypes.data, qb.nbytes), "upload")
    capi.check(h.lib.idocp_device_upload(dv, vb.ctypes.data, vb.nbytes), "upload")
    capi.check(h.lib.idocp_ocp_launch_kernel(h.h, 2, dq, dv), "backward sweep")
    capi.check(h.lib.idocp_ocp_synchronize(h.h), "synchronize")
    h.lib.idocp_device_free(dq)
    h.lib.idocp_device_free(dv)
    worst = 0.0
    for inst in (0, 2):
        Pm, s, K, k = h.riccati(inst)
        for i in range(N + 1):
            r = g["riccati"][i]
            pairs = [(Pm[i], r["P"]), (s[i], r["s"])] + ([(K[i], r["K"]), (k[i], r["k"])] if i < N else [])
            for have, want in pairs:
                want = np.asarray(want)
                err = np.abs(have - want).max() / max(1.0, np.abs(want).max())
                worst = max(worst, err)
                assert err <= 1e-10, (sweep, inst, i, err)
    print("S3 form %d against the dense KKT solution: worst %.2e" % (sweep, worst))
