"""GPU parity tests of the OCPSolver hot path (ANYmal, uniform 4-contact horizon)
through the C ABI.  Bar: Newton direction within 1e-10 (FP64) of the CPU
restatement (BASELINE.json north_star), as max-abs error / max(1, max|ref|)."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipOCP, OracleOCP, P, anymal_contact_points,
                     anymal_model, anymal_problem, arr, rel_err)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def make_pair(N, T, batch=1, trotting_ref=True, perturb=0.02, seed=3, active=(1, 1, 1, 1)):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=trotting_ref)
    o = OracleOCP(m, cost, cons, T, N)
    g = HipOCP(m, cost, cons, T, N, batch=batch)
    pts = anymal_contact_points(m)
    q = ANYMAL_Q_STANDING.copy()
    v = np.zeros(m.nv)
    f0 = [0, 0, 0.25 * (-m.total_mass * m.gravity[2])]
    for s in (o, g):
        s.set_contact_status(active, pts)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", f0)
        s.init_constraints(0.0)
    rng = np.random.default_rng(seed)
    q[7:] += perturb * rng.uniform(-1, 1, 12)
    q[0:2] += perturb * rng.uniform(-1, 1, 2)
    return m, o, g, q, v


def test_lqr_stage_parity_after_linearisation_kernels():
    # kernel-level: K5a + K5b against SplitOCP::linearizeOCP of the oracle
    m, o, g, q, v = make_pair(6, 0.3)
    assert o.update(0.2, q, v) == 0 and g.update(0.2, q, v) == 0       # move off the initial guess
    assert o.stage(0, 0.2, q, v) == 0
    d_q, d_v = C.c_void_p(), C.c_void_p()
    lib = capi.lib()
    capi.check(lib.idocp_device_alloc(C.byref(d_q), q.nbytes))
    capi.check(lib.idocp_device_alloc(C.byref(d_v), v.nbytes))
    capi.check(lib.idocp_device_upload(d_q, arr(q).ctypes.data, q.nbytes))
    capi.check(lib.idocp_device_upload(d_v, arr(v).ctypes.data, v.nbytes))
    # solution states must agree before comparing the linearisation
    for f in OCP_SOL_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    capi.check(lib.idocp_ocp_launch_kernel(g.h, 0, d_q, d_v))
    capi.check(lib.idocp_ocp_launch_kernel(g.h, 1, d_q, d_v))
    capi.check(lib.idocp_ocp_synchronize(g.h))
    names = ("Qxx", "Qxu", "Quu", "A", "B", "lx", "lu", "Fx")
    for i in range(o.N):
        for name, a, b in zip(names, g.lqr_stage(i), o.lqr_stage(i)):
            if name == "Qxx":       # the lower-left block (Qvq) is only filled by the Riccati sweep
                a, b = np.triu(a), np.triu(b)
            assert rel_err(a, b) < TOL, (i, name)


def test_condense_launch_in_two_halves_gives_the_same_records():
    # idocp_ocp_launch_kernel ids 7 + 8 (nominal sweeps | condensation launches, bracketed apart by bench.py) == id 1
    res = []
    for ids in ((0, 1), (0, 7, 8)):
        m, o, g, q, v = make_pair(6, 0.3)
        assert g.update(0.2, q, v) == 0
        d_q, d_v = C.c_void_p(), C.c_void_p()
        lib = capi.lib()
        capi.check(lib.idocp_device_alloc(C.byref(d_q), q.nbytes))
        capi.check(lib.idocp_device_alloc(C.byref(d_v), v.nbytes))
        capi.check(lib.idocp_device_upload(d_q, arr(q).ctypes.data, q.nbytes))
        capi.check(lib.idocp_device_upload(d_v, arr(v).ctypes.data, v.nbytes))
        for kid in ids:
            capi.check(lib.idocp_ocp_launch_kernel(g.h, kid, d_q, d_v))
        capi.check(lib.idocp_ocp_synchronize(g.h))
        res.append([g.lqr_stage(i) for i in range(o.N)])
    for sa, sb in zip(*res):
        for a, b in zip(sa, sb):
            assert np.array_equal(np.triu(a) if a.ndim == 2 and a.shape[0] == a.shape[1] else a, np.triu(b) if b.ndim == 2 and b.shape[0] == b.shape[1] else b)
    assert lib.idocp_ocp_launch_kernel(g.h, 9, d_q, d_v) != 0


@pytest.mark.parametrize("N,T,t", [(20, 1.0, 0.0), (100, 5.0, 0.0), (9, 0.45, 0.37)])
def test_first_iteration_direction_parity(N, T, t):
    m, o, g, q, v = make_pair(N, T)
    assert o.update(t, q, v) == 0
    assert g.update(t, q, v) == 0
    for f in OCP_DIR_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-10 and abs(bg[0] - bo) < 1e-10
    for f in OCP_SOL_FIELDS:
        assert rel_err(g.get(f), o.get(f)) < TOL, f
    Po, so, Ko, ko = o.riccati()
    Pg, sg, Kg, kg = g.riccati()
    assert rel_err(Pg, Po) < TOL and rel_err(sg, so) < TOL and rel_err(Kg, Ko) < TOL and rel_err(kg, ko) < TOL
    sl_o, du_o = o.constraint_data()
    sl_g, du_g = g.constraint_data()
    assert rel_err(sl_g, sl_o) < TOL and rel_err(du_g, du_o) < TOL


def test_multi_iteration_parity_and_convergence():
    m, o, g, q, v = make_pair(20, 0.5, trotting_ref=False)
    e_o = [o.kkt_error(0.0, q, v)]
    e_g = [g.kkt_error(0.0, q, v)[0]]
    assert abs(e_g[0] - e_o[0]) < 1e-9 * max(1.0, e_o[0])
    for it in range(30):
        assert o.update(0.0, q, v) == 0
        assert g.update(0.0, q, v) == 0
        e_o.append(o.kkt_error(0.0, q, v))
        e_g.append(g.kkt_error(0.0, q, v)[0])
        if it < 4:
            # later iterates: rounding differences are amplified by the IPM's ill-conditioning
            # (first-iteration parity at 1e-10 is asserted in test_first_iteration_direction_parity)
            for f in OCP_DIR_FIELDS:
                assert rel_err(g.get(f), o.get(f)) < 1e-6, (it, f)
            assert abs(e_g[-1] - e_o[-1]) < 1e-6 * max(1.0, e_o[-1])
    assert e_g[-1] < 1e-4 * e_g[0] and e_o[-1] < 1e-4 * e_o[0]
    for f in ("q", "v", "a", "u", "f"):
        assert rel_err(g.get(f), o.get(f)) < 1e-6, f


def test_batch_instances_and_ragged_sizes():
    m = anymal_model()
    cost, cons = anymal_problem(m)
    pts = anymal_contact_points(m)
    N, T = 12, 0.6
    f0 = [0, 0, 0.25 * (-m.total_mass * m.gravity[2])]
    rng = np.random.default_rng(20250)
    for batch in (1, 5):
        g = HipOCP(m, cost, cons, T, N, batch=batch)
        g.set_contact_status([1, 1, 1, 1], pts)
        q0 = np.tile(ANYMAL_Q_STANDING, (batch, 1))
        q0[:, 7:] += 0.02 * rng.uniform(-1, 1, (batch, 12))
        q0[:, 0:2] += 0.02 * rng.uniform(-1, 1, (batch, 2))
        v0 = np.zeros((batch, m.nv))
        g.set_solution_batch("q", q0)
        g.set_solution("v", v0[0])
        g.set_solution("f", f0)
        g.init_constraints(0.0)
        assert g.update(0.1, q0, v0) == 0
        for b in range(batch):
            o = OracleOCP(m, cost, cons, T, N)
            o.set_contact_status([1, 1, 1, 1], pts)
            o.set_solution("q", q0[b])
            o.set_solution("v", v0[b])
            o.set_solution("f", f0)
            o.init_constraints(0.0)
            assert o.update(0.1, q0[b], v0[b]) == 0
            for f in OCP_DIR_FIELDS:
                assert rel_err(g.get(f, b), o.get(f)) < TOL, (batch, b, f)


def test_state_feedback_gain():
    m, o, g, q, v = make_pair(8, 0.4)
    o.update(0.0, q, v)
    g.update(0.0, q, v)
    _, _, Ko, _ = o.riccati()
    Kq, Kv = np.zeros((m.nv, m.nu)), np.zeros((m.nv, m.nu))
    capi.check(g.lib.idocp_ocp_get_state_feedback_gain(g.h, 0, 3, P(Kq), P(Kv)))
    assert rel_err(Kq.T, Ko[3][:, :m.nv]) < TOL and rel_err(Kv.T, Ko[3][:, m.nv:]) < TOL
    # ParNMPC handles do not carry the line search (the OCPSolver does since round 2: test_line_search_*)
    assert g.lib.idocp_ocp_update_solution(g.h, 0.0, P(arr(q)), P(arr(v)), 1) == 0


def test_full_size_properties_c3():
    # BASELINE config C3 size (ANYmal, N=100, 4 contacts) with a batch: size-independent properties
    m = anymal_model()
    cost, cons = anymal_problem(m)
    pts = anymal_contact_points(m)
    N, T, batch = 100, 5.0, 64
    g = HipOCP(m, cost, cons, T, N, batch=batch)
    g.set_contact_status([1, 1, 1, 1], pts)
    rng = np.random.default_rng(20250)
    q0 = np.tile(ANYMAL_Q_STANDING, (batch, 1))
    q0[:, 7:] += 0.02 * rng.uniform(-1, 1, (batch, 12))
    q0[1] = q0[0]
    v0 = np.zeros((batch, m.nv))
    g.set_solution_batch("q", q0)
    g.set_solution("v", v0[0])
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    e0 = g.kkt_error(0.0, q0, v0)
    for _ in range(40):
        assert g.update(0.0, q0, v0) == 0
    e1 = g.kkt_error(0.0, q0, v0)
    assert np.isfinite(e1).all() and (e1 < 1e-2 * e0).all(), (e0.max(), e1.max())
    assert np.array_equal(g.get("q", 0), g.get("q", 1))              # identical instances, identical results
    qs = g.get("q", 5)
    assert np.abs(np.linalg.norm(qs[:, 3:7], axis=1) - 1).max() < 1e-12   # quaternions stay normalised
    Pm, s, K, k = g.riccati(5)
    assert np.abs(Pm - Pm.transpose(0, 2, 1)).max() < 1e-8 * np.abs(Pm).max()
    sl, du = g.constraint_data(5)
    assert (sl[2:] > 0).all() and (du[2:] > 0).all()


def test_clone_is_a_deep_copy_of_the_solver_state():
    """idocp_ocp_clone (the copy constructor of the facade's OCPSolver / ParNMPCSolver; the reference's classes are copyable): the
    copy continues exactly like the original, and the two do not share state."""
    import copy
    import ctypes as C
    from idocp_amd import capi
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    g = HipOCP(m, cost, cons, 0.5, 10, batch=2)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    g.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    g.set_solution("q", q)
    g.set_solution("v", v)
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    q[7:] += 0.03
    for _ in range(2):
        assert g.update(0.0, q, v) == 0
    h2 = C.c_void_p()
    capi.check(g.lib.idocp_ocp_clone(g.h, C.byref(h2)), "clone")
    c = copy.copy(g)
    c.h = h2
    before = {f: g.get(f, 1).copy() for f in ("q", "v", "lmd")}
    for _ in range(2):
        assert c.update(0.0, q, v) == 0
    for f in before:                                    # the original did not move
        assert np.array_equal(g.get(f, 1), before[f])
    for _ in range(2):
        assert g.update(0.0, q, v) == 0
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu"):
        assert np.array_equal(g.get(f, 1), c.get(f, 1)), f
    assert np.array_equal(g.kkt_error(0.0, q, v), c.kkt_error(0.0, q, v))


def _ls_pair(N, T, nimp, batch=2, cone="linearized"):
    import ctypes as C
    from helpers import oracle, trotting_sequence
    lib_o = oracle()
    lib_o.oracle_ocp_cost_and_violation.argtypes = [C.c_void_p, C.c_double, capi.c_double_p]
    lib_o.oracle_ocp_compute_direction.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
    lib_o.oracle_ocp_update_solution_ls.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True, cone=cone)
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1 if nimp else 0)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1 if nimp else 0)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        if nimp:
            trotting_sequence(s, m, nimp)
        else:
            s.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    q[7:] += 0.05
    return lib_o, m, o, g, q, v


@pytest.mark.parametrize("cone", ["linearized", "nonlinear"])
@pytest.mark.parametrize("N,T,nimp", [(20, 1.0, 0), (31, 1.55, 2)])
def test_line_search_cost_and_violation_parity(N, T, nimp, cone):
    """Floating-base filter line search (src/line_search/line_search.cpp:63-196): total cost and l1 constraint violation of the
    trial iterates s (+) alpha d against the oracle, on an event-free horizon and on the trotting chain (impulse / aux / lift
    stages, switching constraints, the reference's pairing of the stages in front of an event)."""
    lib_o, m, o, g, q, v = _ls_pair(N, T, nimp, cone=cone)          # (nonlinear: FrictionCone / ImpulseFrictionCone rows in the barrier and the violation)
    assert lib_o.oracle_ocp_compute_direction(o.h, 0.0, P(q), P(v)) == 0
    capi.check(g.lib.idocp_ocp_compute_direction(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))), "compute_direction")
    ap, ad = g.step_sizes()
    for alpha in (0.0, 1e-3, 0.1, 0.5 * ap[0], ap[0]):
        ref = np.zeros(2)
        assert lib_o.oracle_ocp_cost_and_violation(o.h, alpha, P(ref)) == 0
        c, vi = np.zeros(g.batch), np.zeros(g.batch)
        capi.check(g.lib.idocp_ocp_line_search_eval(g.h, P(np.full(g.batch, alpha)), P(c), P(vi)), "line_search_eval")
        assert abs(c[0] - ref[0]) <= 1e-10 * max(1.0, abs(ref[0])), (alpha, c[0], ref[0])
        assert abs(vi[0] - ref[1]) <= 1e-10 * max(1.0, abs(ref[1])), (alpha, vi[0], ref[1])
        assert c[0] == c[-1] and vi[0] == vi[-1]


def test_line_search_accepted_steps_follow_the_oracle():
    """updateSolution(t, q, v, line_search = true) (ocp_solver.cpp:84-90): the accepted primal steps and the iterates of GPU and
    oracle over several iterations (one filter per instance), then clearLineSearchFilter."""
    lib_o, m, o, g, q, v = _ls_pair(20, 1.0, 0)
    for it in range(6):
        assert lib_o.oracle_ocp_update_solution_ls(o.h, 0.0, P(q), P(v)) == 0
        assert g.lib.idocp_ocp_update_solution(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv)), 1) == 0
        ao, bo = o.step_sizes()
        ag, bg = g.step_sizes()
        assert abs(ag[0] - ao) < 1e-12 and abs(bg[0] - bo) < 1e-9, (it, ag[0], ao)
        for f in ("q", "v", "a", "u", "f"):
            assert rel_err(g.get(f, 1), o.get(f)) < 1e-8, (it, f)
    capi.check(g.lib.idocp_ocp_clear_line_search_filter(g.h))
