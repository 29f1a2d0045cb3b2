"""GPU parity tests of the UnOCPSolver hot path (iiwa14), through the C ABI.

Bar (BASELINE.json north_star): the Newton direction matches the CPU restatement
to 1e-10 in FP64.  TOL below is that tolerance, applied as max-abs error divided
by max(1, max|reference|) per field."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from helpers import (DIR_FIELDS, SOL_FIELDS, HipUnOCP, OracleUnOCP, P, arr, iiwa14_model, load_golden, oracle, rel_err,
                     unocp_problem)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def make_pair(N, T, batch=1):
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnOCP(m, cost, cons, T, N)
    g = HipUnOCP(m, cost, cons, T, N, batch=batch)
    q = np.full(m.nv, 2.0)
    v = np.zeros(m.nv)
    for s in (o, g):
        s.set_solution("q", q)
        s.set_solution("v", v)
    return m, o, g, q, v


def test_extension_loaded_and_gpu_present():
    n = C.c_int()
    capi.lib().idocp_device_count(C.byref(n))
    assert n.value >= 1, "no HIP device visible: the product path has no CPU fallback"


def test_rnea_derivatives_vs_golden_and_oracle():
    g = load_golden("iiwa14")
    m = iiwa14_model()
    nv = m.nv
    rng = np.random.default_rng(7)
    n = 64 + len(g["samples"])
    q = np.vstack([arr([s["q"] for s in g["samples"]]), rng.uniform(-2.5, 2.5, (64, nv))])
    v = np.vstack([arr([s["v"] for s in g["samples"]]), rng.uniform(-3, 3, (64, nv))])
    a = np.vstack([arr([s["a"] for s in g["samples"]]), rng.uniform(-5, 5, (64, nv))])
    q, v, a = arr(q), arr(v), arr(a)
    tau, dq, dv, da = np.zeros((n, nv)), np.zeros((n, nv, nv)), np.zeros((n, nv, nv)), np.zeros((n, nv, nv))
    capi.check(capi.lib().idocp_rnea_derivatives(C.byref(m), n, P(q), P(v), P(a), P(tau), P(dq), P(dv), P(da), 0))
    for i, s in enumerate(g["samples"]):          # committed golden vectors
        assert rel_err(tau[i], s["tau"]) < 1e-12
        assert rel_err(dq[i].T, s["dtau_dq"]) < 1e-12
        assert rel_err(dv[i].T, s["dtau_dv"]) < 1e-12
        assert rel_err(da[i].T, s["dtau_da"]) < 1e-12
    ol = oracle()
    t0, d0, d1, d2 = np.zeros(nv), np.zeros((nv, nv)), np.zeros((nv, nv)), np.zeros((nv, nv))
    for i in range(n):                            # oracle on the same seeded inputs
        ol.oracle_rnea(C.byref(m), P(q[i]), P(v[i]), P(a[i]), None, 1, P(t0))
        ol.oracle_rnea_derivatives(C.byref(m), P(q[i]), P(v[i]), P(a[i]), None, 1, P(d0), P(d1), P(d2))
        assert rel_err(tau[i], t0) < 1e-12
        assert rel_err(dq[i], d0) < 1e-12 and rel_err(dv[i], d1) < 1e-12 and rel_err(da[i], d2) < 1e-12


@pytest.mark.parametrize("N,T", [(20, 1.0), (100, 5.0), (7, 0.35)])
def test_first_iteration_direction_parity(N, T):
    m, o, g, q, v = make_pair(N, T)
    assert o.update(0.0, q, v) == 0
    assert g.update(0.0, q, v) == 0
    for f in DIR_FIELDS:
        assert rel_err(g.direction(f), o.direction(f)) < TOL, f
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-10 and abs(bg[0] - bo) < 1e-10
    for f in SOL_FIELDS:
        assert rel_err(g.solution(f), o.solution(f)) < TOL, f


def test_riccati_factorization_parity():
    m, o, g, q, v = make_pair(20, 1.0)
    o.update(0.0, q, v)
    g.update(0.0, q, v)
    Po, so, Ko, ko = o.riccati()
    Pg, sg, Kg, kg = g.riccati()
    assert rel_err(Pg, Po) < TOL and rel_err(sg, so) < TOL
    assert rel_err(Kg, Ko) < TOL and rel_err(kg, ko) < TOL


def test_multi_iteration_parity_and_convergence():
    # ocpbenchmarker::Convergence protocol (ocp_benchmarker.hxx:37-52), config C1
    m, o, g, q, v = make_pair(20, 1.0)
    e_o = [o.kkt_error(0.0, q, v)]
    e_g = [g.kkt_error(0.0, q, v)[0]]
    assert abs(e_g[0] - e_o[0]) < 1e-9 * max(1.0, e_o[0])
    for it in range(50):       # num_iteration = 50 in examples/iiwa14/unocp_benchmark.cpp:50
        assert o.update(0.0, q, v) == 0
        assert g.update(0.0, q, v) == 0
        e_o.append(o.kkt_error(0.0, q, v))
        e_g.append(g.kkt_error(0.0, q, v)[0])
        if it < 5:       # iterate-by-iterate parity while the iterates are still far from the optimum
            for f in DIR_FIELDS:
                assert rel_err(g.direction(f), o.direction(f)) < 1e-8, (it, f)
            assert abs(e_g[-1] - e_o[-1]) < 1e-8 * max(1.0, e_o[-1])
    assert e_g[-1] < 1e-6 * e_g[0] and e_o[-1] < 1e-6 * e_o[0]
    # the IPM iterates are sensitive to rounding far from the optimum, but both paths
    # must land on the same optimum
    for f in ("q", "v", "a", "u"):
        assert rel_err(g.solution(f), o.solution(f)) < 1e-6, f
    sl_o, du_o = o.constraint_data()
    sl_g, du_g = g.constraint_data()
    assert rel_err(sl_g, sl_o) < 1e-6 and rel_err(du_g, du_o) < 1e-6


def test_initial_state_offset_and_slack_data():
    # the measured state differs from s[0]: d[0] = (q - s0.q, v - s0.v) (unocp_solver.cpp:100-101)
    m, o, g, q, v = make_pair(20, 1.0)
    q2, v2 = q + np.linspace(-0.1, 0.1, m.nv), v + 0.3
    o.update(0.0, q2, v2)
    g.update(0.0, q2, v2)
    assert rel_err(g.direction("dq")[0], q2 - q) < 1e-14
    for f in DIR_FIELDS:
        assert rel_err(g.direction(f), o.direction(f)) < TOL, f
    sl_o, du_o = o.constraint_data()
    sl_g, du_g = g.constraint_data()
    assert rel_err(sl_g, sl_o) < TOL and rel_err(du_g, du_o) < TOL


def test_batch_instances_are_independent_and_ragged_batch_sizes():
    # batch sizes that do not fill the last wavefront (8 instances / wave in S1, 9 consecutive stages / wave in K1: with N = 20 the
    # boundaries between instances fall inside wavefronts; N = 4 and N = 1 put two and nine of them into one)
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    rng = np.random.default_rng(20240)
    for N, T, batch in ((20, 1.0, 1), (20, 1.0, 3), (20, 1.0, 11), (4, 0.2, 7), (1, 0.05, 10)):
        g = HipUnOCP(m, cost, cons, T, N, batch=batch)
        q0 = 2.0 + 0.1 * rng.uniform(-1, 1, (batch, m.nv))
        v0 = np.zeros((batch, m.nv))
        g.set_solution_batch("q", q0)
        g.set_solution("v", v0[0])
        assert g.update(0.0, q0, v0) == 0
        for b in range(batch):
            o = OracleUnOCP(m, cost, cons, T, N)
            o.set_solution("q", q0[b])
            o.set_solution("v", v0[b])
            o.update(0.0, q0[b], v0[b])
            for f in DIR_FIELDS:
                assert rel_err(g.direction(f, b), o.direction(f)) < TOL, (N, batch, b, f)


def test_full_size_properties_c2():
    # BASELINE config C2 size (N=100) with a large batch: size-independent properties
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    N, T, batch = 100, 5.0, 512
    g = HipUnOCP(m, cost, cons, T, N, batch=batch)
    rng = np.random.default_rng(20240)
    q0 = 2.0 + 0.1 * rng.uniform(-1, 1, (batch, m.nv))
    q0[1] = q0[0]                                  # identical instances must give identical results
    v0 = np.zeros((batch, m.nv))
    g.set_solution_batch("q", q0)
    g.set_solution("v", v0[0])
    e0 = g.kkt_error(0.0, q0, v0)
    for _ in range(80):
        assert g.update(0.0, q0, v0) == 0
    e1 = g.kkt_error(0.0, q0, v0)
    assert np.isfinite(e1).all() and (e1 < 1e-3 * e0).all(), (e0.max(), e1.max())
    assert np.array_equal(g.solution("q", 0), g.solution("q", 1))
    dt = T / N
    for b in (0, 7, batch - 1):
        Pm, s, K, k = g.riccati(b)
        assert np.abs(Pm - Pm.transpose(0, 2, 1)).max() < 1e-9 * np.abs(Pm).max()      # P symmetric
        dq, dv, da = g.direction("dq", b), g.direction("dv", b), g.direction("da", b)
        # the direction satisfies the linearised state equation: residuals after a full step vanish
        qs, vs, as_ = g.solution("q", b), g.solution("v", b), g.solution("a", b)
        ap, _ = g.step_sizes()
        Fq = qs[:-1] - qs[1:] + dt * vs[:-1]
        Fv = vs[:-1] + dt * as_ - vs[1:]
        if ap[b] == 1.0:
            assert np.abs(Fq).max() < 1e-8 and np.abs(Fv).max() < 1e-8
        sl, du = g.constraint_data(b)
        assert (sl[2:] > 0).all() and (du[2:] > 0).all()


def test_filter_line_search_parity():
    # UnLineSearch (include/idocp/line_search/unline_search.hpp:62-92): the trial-iterate cost / violation and the accepted
    # step of every iteration against the oracle
    m, o, g, q, v = make_pair(20, 1.0, batch=2)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    # the state the line search sees: direction and step sizes computed, iterate not yet updated
    for what in (0, 1, 2):
        assert o.stage(what, 0.0, q, v) == 0
    for kid in (0, 1, 2, 3, 4):
        g.launch(kid, q, v)
    amax = o.step_sizes()[0]
    for alpha in (0.0, 0.5 * amax, amax):
        co, vo = o.cost_and_violation(alpha)
        cg, vg = g.cost_and_violation(alpha)
        assert abs(cg[0] - co) < 1e-10 * max(1.0, abs(co)) and abs(vg[0] - vo) < 1e-10 * max(1.0, vo), (alpha, cg, co, vg, vo)
        assert cg[0] == cg[1] and vg[0] == vg[1]
    assert o.stage(3, 0.0, q, v) == 0
    g.launch(5, q, v)
    for it in range(8):
        assert o.update(0.0, q, v, line_search=True) == 0 and g.update(0.0, q, v, line_search=True) == 0
        ao, _ = o.step_sizes()
        ag, _ = g.step_sizes()
        assert abs(ag[0] - ao) < 1e-9 and ag[0] == ag[1], (it, ag, ao)
        for f in ("q", "v", "a", "u"):
            assert rel_err(g.solution(f), o.solution(f)) < 1e-8, (it, f)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) < 1e-7 * max(1.0, e_o)
    # clearLineSearchFilter: the next call starts from an empty filter again on both sides
    o.clear_line_search_filter()
    g.clear_line_search_filter()
    assert o.update(0.0, q, v, line_search=True) == 0 and g.update(0.0, q, v, line_search=True) == 0
    assert abs(g.step_sizes()[0][0] - o.step_sizes()[0]) < 1e-9


def test_general_axes_instantiation_direction_parity(monkeypatch):
    # iiwa14 qualifies for the K1 instantiation that knows at compile time that every joint turns about +z (UnBuffers::zaxes,
    # dev_rbd.hpp rneaChain<NJ, ZAX>); IDOCP_GENERAL_AXES forces the general one, which any other 7-joint chain would run
    monkeypatch.setenv("IDOCP_GENERAL_AXES", "1")
    m, o, g_general, q, v = make_pair(20, 1.0)
    monkeypatch.delenv("IDOCP_GENERAL_AXES")
    _, _, g_special, _, _ = make_pair(20, 1.0)
    assert o.update(0.0, q, v) == 0 and g_general.update(0.0, q, v) == 0 and g_special.update(0.0, q, v) == 0
    for f in DIR_FIELDS:
        assert rel_err(g_general.direction(f), o.direction(f)) < TOL, f
        assert rel_err(g_special.direction(f), o.direction(f)) < TOL, f
        assert rel_err(g_general.direction(f), g_special.direction(f)) < TOL, f



def test_set_cost_moves_the_goal_like_the_shared_cost_function_of_the_reference():
    """The reference's UnOCPSolver / UnParNMPCSolver share the CostFunction with the driver (shared_ptr members, unocp_solver.hpp): an MPC loop that
    moves q_ref or re-weights between two updates sees it at the next one.  idocp_unocp_set_cost is that on a handle: from the same iterate the
    step equals, bit for bit, the step of a solver CREATED with the new cost, and the oracle's for that cost to 1e-10."""
    from helpers import HipUnParNMPC, OracleUnParNMPC
    m = iiwa14_model()
    cost_a, cons = unocp_problem(m)
    cost_b, _ = unocp_problem(m)
    cost_b.set("q_ref", np.linspace(-1.0, 1.0, m.nv)).set("v_ref", np.zeros(m.nv))
    cost_b.set("q_weight", np.full(m.nv, 3.0)).set("qf_weight", np.full(m.nv, 7.0)).set("u_weight", np.full(m.nv, 1e-3))
    T, N = 1.0, 20
    q, v = np.full(m.nv, 0.4), np.zeros(m.nv)
    lib = capi.lib()
    for Hip, Orc in ((HipUnOCP, OracleUnOCP), (HipUnParNMPC, OracleUnParNMPC)):
        un = Hip is HipUnOCP
        moved, fresh, o = Hip(m, cost_a, cons, T, N), Hip(m, cost_b, cons, T, N), Orc(m, cost_b, cons, T, N)
        capi.check(lib.idocp_unocp_set_cost(moved.h, C.byref(cost_b)), "set_cost")
        for s in (moved, fresh, o):
            s.set_solution("q", q)
            s.set_solution("v", v)
            if not un:
                s.init(0.0)                  # (the aux matrices of initBackwardCorrection carry the terminal weights)
        e_m, e_f = moved.kkt_error(0.0, q, v)[0], fresh.kkt_error(0.0, q, v)[0]
        assert e_m == e_f and abs(e_m - o.kkt_error(0.0, q, v)) <= 1e-10 * e_f
        for it in range(2):
            assert moved.update(0.0, q, v) == 0 and fresh.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
            for name in DIR_FIELDS:
                a, b, r = (moved.direction(name), fresh.direction(name), o.direction(name)) if un else (moved.get(name), fresh.get(name), o.get(name))
                assert np.array_equal(a, b), (it, name)
                assert rel_err(a, r) <= (TOL if it == 0 else 1e-8), (it, name, rel_err(a, r))
    # the kind of cost is fixed at creation
    cost_t, _ = unocp_problem(m)
    cost_t.task_dim = 3
    g = HipUnOCP(m, cost_a, cons, T, N)
    assert lib.idocp_unocp_set_cost(g.h, C.byref(cost_t)) != 0
    assert b"cannot be added or removed" in lib.idocp_last_error()
