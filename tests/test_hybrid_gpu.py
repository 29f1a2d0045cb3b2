"""GPU parity of the hybrid (discrete-event) OCPSolver path against the oracle, stage by stage along
the chain (stage, [impulse, aux | lift], ..., terminal), on the reference's trotting problem
(examples/anymal/anymal_trotting.cpp transcribed as data).  Bar: 1e-10 on the Newton direction (FP64)."""
import numpy as np
import pytest

from idocp_amd import capi

from helpers import (parity, pairwise_check, ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipOCP, OracleOCP, anymal_model, anymal_problem, referee_check,
                     rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def make_pair(N, T, nimp, batch=1, lift_only=False, referee=False):
    """GPU solver and FP64 oracle on the trotting problem; referee=True adds the long double build of the oracle (returned last)."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True) if referee else None
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g) + ((h,) if referee else ()):
        trotting_sequence(s, m, 0 if lift_only else nimp)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    return (m, o, g, q, v, h) if referee else (m, o, g, q, v)


def compare_chain(o, g, M, fields, tol, what):
    worst = 0.0
    for f in fields:
        e = rel_err(g.get_chain(f, M), o.get_chain(f, M))
        worst = max(worst, e)
        assert e < tol, (what, f, e)
    return worst


def test_chain_matches_the_oracle_discretiser():
    m, o, g, q, v = make_pair(30, 1.55, 2)
    for t in (0.0, 0.23):
        co, cg = o.chain(t), g.chain(t)
        assert len(co) == len(cg)
        for a, b in zip(co, cg):
            assert a["kind"] == b["kind"] and a["index"] == b["index"] and a["slot"] == b["slot"] and a["dimf"] == b["dimf"]
            assert abs(a["dt"] - b["dt"]) < 1e-15
            assert (a["sw_event"] >= 0) == (b["sw_dimi"] > 0)


# (N, T, tolerance).  The example's own discretisation (N = 30, T = 1.55) puts a 1.7 ms stage right before the second
# impulse: the switching-constraint Schur complement of that stage drives |P| to 7e6 (vs 1e4 elsewhere) and FP64
# rounding alone separates ANY two evaluation orders there.  tolerance None = decided by the long double referee: the GPU
# may be at most 4x as far from it as the FP64 oracle is (+ 1e-10), stage by stage; the other grids are conditioned
# normally and must meet the plain 1e-10 bar against the oracle.
GRIDS = [(31, 1.55, TOL), (30, 1.6, TOL), (32, 1.55, TOL), (40, 1.8, TOL), (30, 1.55, None)]


@pytest.mark.parametrize("N,T,tol", GRIDS)
@pytest.mark.parametrize("lift_only", [True, False])
def test_first_iteration_direction_parity_along_the_chain(lift_only, N, T, tol):
    if lift_only and (N, T) != (30, 1.55):
        pytest.skip("lift-only sequence: one grid is enough")
    use_referee = tol is None and not lift_only
    m, o, g, q, v, h = make_pair(N, T, 2, lift_only=lift_only, referee=True)
    qq = q.copy()
    qq[7:] += 0.02 * np.random.default_rng(4).uniform(-1, 1, 12)
    assert o.update(0.0, qq, v) == 0
    assert g.update(0.0, qq, v) == 0
    assert h.update(0.0, qq, v) == 0
    M = len(o.chain(0.0))
    dirs = list(OCP_DIR_FIELDS) + ([] if lift_only else ["dxi"])
    for f in dirs:       # every grid: never further from the long double referee than 4x the FP64 oracle + 1e-10
        referee_check(g.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
    TOL = 1e-8 if use_referee else 1e-10     # against the oracle itself: the plain bar (a loose cap where the referee decides)
    compare_chain(o, g, M, dirs, TOL, "direction")
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-10 and abs(bg[0] - bo) < 1e-10
    compare_chain(o, g, M, list(OCP_SOL_FIELDS) + ["xi"], TOL, "solution")
    Po, so, Ko, ko = o.riccati_chain(M)
    Pg, sg, Kg, kg = g.riccati_chain(M)
    assert rel_err(Pg, Po) < TOL and rel_err(sg, so) < TOL and rel_err(Kg, Ko) < TOL and rel_err(kg, ko) < TOL


def test_hybrid_convergence_and_kkt_error_parity():
    m, o, g, q, v = make_pair(30, 1.55, 2, batch=3)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-9 * max(1.0, e_o)
    M = len(o.chain(0.0))
    for it in range(25):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
        assert np.allclose(e_g, e_g[0], rtol=0, atol=1e-9 * max(1.0, e_o))          # the three instances are identical problems
        tol = 1e-8 if it == 0 else 1e-6                                              # see GRIDS; rounding differences amplify along the SQP path
        assert abs(e_g[0] - e_o) <= tol * max(1.0, e_o) + 1e-10, (it, e_g[0], e_o)
    assert e_g[0] < 1e-8
    compare_chain(o, g, M, ["q", "v", "a", "u", "f"], 1e-6, "converged solution")


def test_moving_horizon_rediscretisation_parity():
    """MPC use: the initial time advances, the events slide across the grid and the chain changes shape while the
    stage records stay in their slots (OCPSolver::updateSolution re-discretises on every call, ocp_solver.cpp:72-73).
    GPU and oracle must follow the same path through several re-discretisations."""
    m, o, g, q, v = make_pair(31, 1.55, 2)
    kinds_seen = set()
    for it, t in enumerate([0.0, 0.0, 0.013, 0.027, 0.05, 0.05, 0.081, 0.11]):
        assert o.update(t, q, v) == 0 and g.update(t, q, v) == 0
        co, cg = o.chain(t), g.chain(t)
        assert [(a["kind"], a["slot"]) for a in co] == [(b["kind"], b["slot"]) for b in cg]
        kinds_seen.add("".join(c["kind"][0] for c in co))
        M = len(co)
        tol = 1e-10 if it == 0 else 1e-6
        for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu"):
            assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < tol, (it, t, f)
        e_o, e_g = o.kkt_error(t, q, v), g.kkt_error(t, q, v)
        assert abs(e_g[0] - e_o) <= 1e-6 * max(1.0, e_o)
    assert len(kinds_seen) >= 2          # the chain did change shape along the way


def test_flight_phase_sequence_parity():
    """A running-style contact sequence with a FLIGHT phase (no active contact: dimf = 0 stages, two lift events in a row,
    then two impulse events), the pattern of examples/anymal/anymal_running.cpp:148-215.  First iterations only: the point
    is that every stage kind / contact dimension combination takes the same path on the GPU and in the oracle."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    N, T, E = 24, 1.0, 5
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    g = HipOCP(m, cost, cons, T, N, max_num_impulse=E)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)
    from helpers import anymal_contact_points
    pts = anymal_contact_points(m).copy()
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g, h):
        s.set_contact_status([1, 1, 1, 1], pts)
        s.push_back_contact_status([0, 1, 0, 1], pts, 0.21)          # front feet lift
        s.push_back_contact_status([0, 0, 0, 0], pts, 0.33)          # flight
        p2 = pts.copy()
        p2[:, 0] += 0.05
        s.push_back_contact_status([1, 0, 1, 0], p2, 0.47)           # front feet touch down (impulse)
        s.push_back_contact_status([1, 1, 1, 1], p2, 0.61)           # hind feet touch down (impulse)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    co, cg = o.chain(0.0), g.chain(0.0)
    assert [(a["kind"], a["slot"], a["dimf"]) for a in co] == [(b["kind"], b["slot"], b["dimf"]) for b in cg]
    assert any(c["dimf"] == 0 and c["kind"] == "stage" for c in co) and sum(1 for c in co if c["kind"] == "impulse") == 2
    M = len(co)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    for f in list(OCP_DIR_FIELDS) + ["dxi"]:           # 1e-10 against the oracle, else the referee rule; never past 2e-8
        parity(g.get_chain(f, M), o.get_chain(f, M), lambda: h.get_chain(f, M), f, cap=2e-8)
    for it in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in ("q", "v", "a", "u", "f"):
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-6, f


def test_odd_contact_counts_take_the_general_class():
    """Stages with THREE feet and with ONE foot in contact (dimf = 9, 3), reached by single-foot lifts and single-foot / two-feet impulses:
    neither 12, 6 nor 0 contact rows, so every such stage -- plain, impulse, with a switching constraint -- runs the general instantiation of
    the condensation kernel (run-time contact count in the wide LDS layout; its contact Schur complement next to the mass-matrix inverse,
    DESIGN 4.0a).  First direction under the referee rule, then a few iterations."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    N, T, E = 28, 1.2, 6
    mk = lambda cls, **kw: cls(m, cost, cons, T, N, max_num_impulse=E, **kw)
    o, g, h = mk(OracleOCP), mk(HipOCP), mk(OracleOCP, hp=True)
    from helpers import anymal_contact_points
    pts = anymal_contact_points(m).copy()
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g, h):
        s.set_contact_status([1, 1, 1, 1], pts)
        s.push_back_contact_status([0, 1, 1, 1], pts, 0.17)          # one foot lifts: three feet, 9 rows
        s.push_back_contact_status([0, 1, 0, 0], pts, 0.36)          # two more: one foot, 3 rows
        p2 = pts.copy()
        p2[:, 0] += 0.03
        s.push_back_contact_status([1, 1, 0, 0], p2, 0.58)           # impulse of one foot (3 impulse rows), two feet after it
        s.push_back_contact_status([1, 1, 1, 0], p2, 0.79)           # impulse of one foot, three feet after it
        s.push_back_contact_status([1, 1, 1, 1], p2, 1.01)           # impulse of the last foot
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    co, cg = o.chain(0.0), g.chain(0.0)
    assert [(a["kind"], a["slot"], a["dimf"]) for a in co] == [(b["kind"], b["slot"], b["dimf"]) for b in cg]
    dims = {c["dimf"] for c in co if c["kind"] == "stage"}
    assert {3, 9} <= dims and sum(1 for c in co if c["kind"] == "impulse") == 3
    M = len(co)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    for f in list(OCP_DIR_FIELDS) + ["dxi"]:
        referee_check(g.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-8, f
    for it in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in ("q", "v", "a", "u", "f"):
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-6, f


def test_running_example_parity():
    """BASELINE.json configs[4]'s problem: examples/anymal/anymal_running.cpp (TimeVaryingConfigurationSpaceCost, 40 discrete
    events -- 26 touch-downs, 14 lift-offs --, flight phases, N = 240, T = 7) on the GPU against the oracle: same chain, the
    first Newton direction under the referee rule (stage by stage), the same KKT error along the first iterations."""
    from helpers import ANYMAL_Q_RUNNING_START, running_problem, running_sequence
    m = anymal_model()
    steps = 10
    cost, cons = running_problem(m, steps)
    N, T, E = 240, 7.0, (steps + 3) * 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)      # long double referee
    g = HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=E)
    q, v = ANYMAL_Q_RUNNING_START.copy(), np.zeros(m.nv)
    for s in (o, g, h):
        assert running_sequence(s, m, steps) == 40
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    co, cg = o.chain(0.0), g.chain(0.0)
    assert [(a["kind"], a["slot"], a["dimf"]) for a in co] == [(b["kind"], b["slot"], b["dimf"]) for b in cg]
    assert max(abs(a["dt"] - b["dt"]) for a, b in zip(co, cg)) < 1e-15
    M = len(co)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-9 * e_o and abs(e_g[1] - e_o) <= 1e-9 * e_o
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    # stage by stage (helpers.rel_err).  Stages of a few milliseconds sit in front of the impulses of this gait: there two FP64
    # evaluations separate by 1e-9 of the stage's own entries, and the long double referee decides (GPU at most 4x as far from it as
    # the FP64 oracle, + 1e-10); 1e-8 is the cap against the oracle.
    for f in list(OCP_DIR_FIELDS) + ["dxi"]:
        referee_check(g.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
        assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-8, f
    for it in range(12):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-5 * e_o, (e_g, e_o)


def test_configs4_grid_and_fp32_storage(capsys):
    """BASELINE.json configs[4] on its own grid: the running gait of examples/anymal/anymal_running.cpp with N = 200 and the
    example's time step (T = 7 * 200 / 240), FP64 against the oracle, and the FP32 TOLERANCE STUDY ON THE DEVICE:
    idocp_ocp_set_riccati_storage(32) rounds the cost-to-go P, s to single precision after every stage of the backward sweep (what an
    FP32 ric record / FP32 copy in S3's LDS would hold; all arithmetic stays FP64).  Along the SQP iterates of the FP64 solver the
    Newton direction of a copy of the solver with FP32 storage is compared with the FP64 direction from the same iterate.
    tests/study_fp32_riccati.py (CPU, numpy) predicted 1e-6 .. 2e-6 for this variant on a uniform all-contact horizon with this cost;
    the table this test prints is the measurement on the gait's own chain."""
    import ctypes as C
    from helpers import ANYMAL_Q_RUNNING_START, P, running_problem, running_sequence
    from idocp_amd import capi
    m = anymal_model()
    steps = 10
    cost, cons = running_problem(m, steps)
    N, T, E = 200, 7.0 * 200 / 240, (steps + 3) * 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)      # long double referee
    g = HipOCP(m, cost, cons, T, N, batch=1, max_num_impulse=E)
    q, v = ANYMAL_Q_RUNNING_START.copy(), np.zeros(m.nv)
    for s in (o, g, h):
        assert running_sequence(s, m, steps) == 40
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    co, cg = o.chain(0.0), g.chain(0.0)
    assert [(a["kind"], a["slot"], a["dimf"]) for a in co] == [(b["kind"], b["slot"], b["dimf"]) for b in cg]
    M = len(co)
    assert sum(1 for a in co if a["kind"] == "stage") == N and M > N + 20      # the grid of the config, with most of the gait's events inside
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-9 * e_o
    lib = g.lib
    dirs = list(OCP_DIR_FIELDS) + ["dxi"]
    rows = []
    for it in range(16):
        # the same iterate, FP64 storage (g) and FP32 storage of P, s (a copy of g)
        h32 = C.c_void_p()
        capi.check(lib.idocp_ocp_clone(g.h, C.byref(h32)), "clone")
        capi.check(lib.idocp_ocp_set_riccati_storage(h32, 32), "set_riccati_storage")
        g32 = HipOCP.__new__(HipOCP)
        g32.__dict__.update(g.__dict__)
        g32.h = h32
        capi.check(lib.idocp_ocp_compute_direction(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))), "compute_direction")
        capi.check(lib.idocp_ocp_compute_direction(h32, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))), "compute_direction (FP32 storage)")
        prim = ("dq", "dv", "da", "du", "df")      # what compute_direction leaves (the dual directions come with the integration kernel)
        d64 = {f: g.get_chain(f, M) for f in prim}
        d32 = {f: g32.get_chain(f, M) for f in prim}
        num = max(np.abs(d32[f] - d64[f]).max() for f in ("dq", "dv", "du"))
        den = max(np.abs(d64[f]).max() for f in ("dq", "dv", "du"))
        rows.append((it, float(g.kkt_error(0.0, q, v)[0]), num / max(den, 1e-300), max(rel_err(d32[f], d64[f]) for f in prim)))
        g32.h = None
        lib.idocp_ocp_destroy(h32)
        assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
        if it == 0:
            assert h.update(0.0, q, v) == 0
            for f in dirs:      # FP64 parity on this grid, stage by stage: referee rule + 1e-8 cap (see test_running_example_parity)
                referee_check(g.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
                assert rel_err(g.get_chain(f, M), o.get_chain(f, M)) < 1e-8, f
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-5 * e_o, (e_g, e_o)
    with capsys.disabled():
        print("\nconfigs[4] (running gait, N = 200): direction error of FP32-stored P, s against FP64, per SQP iterate")
        print("iterate   KKT error    |d32 - d64| / |d64| (dx, du; whole horizon)   worst stage and field (helpers.rel_err)")
        for it, kkt, e_all, e_stage in rows:
            print("%5d   %10.3e   %10.2e   %10.2e" % (it, kkt, e_all, e_stage))
    worst = max(r[2] for r in rows)
    # FP32 storage is visible -- and two to three orders of magnitude MORE visible than the CPU study predicted: that study swept the
    # stage blocks of a uniform all-contact horizon (1e-6 .. 2e-6); on the gait's own chain, with flight phases, impulse stages and
    # switching constraints, the direction moves by 1e-4 .. 1e-3 of its size (1e-2 on single stages).  FP64 stays the product.
    assert 1e-9 < worst < 1e-1, worst


def test_full_size_c3_trotting_parity_and_properties():
    # BASELINE configs[2] at its own size, the default bench workload: N = 100, T = 5.05, 1 lift + 9 impulse events
    # (120 stages in the chain).  First iterations against the oracle along the whole chain, then size-independent properties.
    nimp = 9
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    T, N = 0.5 + nimp * 0.5 + 0.05, 100
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)      # long double referee
    g = HipOCP(m, cost, cons, T, N, batch=3, max_num_impulse=nimp + 1)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g, h):
        trotting_sequence(s, m, nimp)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    M = len(o.chain(0.0))
    assert M == 120 and len(g.chain(0.0)) == M
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-9 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    # The last touch-down of this schedule sits 5 ms after a grid point: on the stage behind that switching constraint
    # G = Quu + B^T P B has a condition number of 3e8 and the FP64 oracle itself is 5e-9 away from its long double build THERE
    # (every stage measured against its own largest entry, helpers.rel_err; 1e-13 on the first half of the chain; round 2 quoted
    # 3.4e-10, with the whole horizon's largest entry as the scale).  The bar is the referee's: stage by stage the GPU is at most 4x as
    # far from the long double result as the FP64 oracle is, + 1e-10 (round 1, with explicit Gauss-Jordan inverses in the Riccati
    # step, was 1e-6 away).  Against the oracle itself: 1e-10 on the first half of the chain, 2e-8 overall.
    worst_g = worst_o = 0.0
    for f in OCP_DIR_FIELDS:
        eg, eo = referee_check(g.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
        worst_g, worst_o = max(worst_g, eg), max(worst_o, eo)
    assert worst_g < 2e-8 and worst_o < 2e-8, (worst_g, worst_o)
    compare_chain(o, g, M, list(OCP_DIR_FIELDS), 2e-8, "first iteration, full size")
    for f in OCP_DIR_FIELDS:
        a, b = np.asarray(g.get_chain(f, M)), np.asarray(o.get_chain(f, M))
        assert rel_err(a[:60], b[:60]) < 1e-10, f
    for it in range(9):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-4 * max(1.0, e_o) and e_g[0] == e_g[2]
    qs = g.get_chain("q", M, 1)
    assert np.abs(np.linalg.norm(qs[:, 3:7], axis=1) - 1).max() < 1e-12
    # THE CONVERGED SOLUTION against the referee's: the three solvers iterate on until the KKT error has stalled (GPU and FP64
    # oracle at their rounding floors), then the GPU's primal solution may be at most 4x as far from the long double solver's as the
    # FP64 oracle's is, + 1e-8 (a solution is only determined to cond(KKT) eps), stage by stage.
    for it in range(30):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
    e_o, e_g, e_h = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0], h.kkt_error(0.0, q, v)
    assert e_g < 1e-6 and e_o < 1e-6 and e_h < 1e-6, (e_g, e_o, e_h)
    for f in ("q", "v", "a", "u", "f"):
        sg, so, sh = (np.asarray(x.get_chain(f, M)) for x in (g, o, h))
        referee_check(sg, so, sh, "converged " + f, tol=1e-8)


def test_general_axes_instantiations_direction_parity(monkeypatch):
    """ANYmal qualifies for the instantiations of K5 that know the joint axes of the legs at compile time (OcpBuffers::leg_axes_xyy,
    dev_rnea_tangent.hpp JointFrame).  IDOCP_GENERAL_AXES forces the general instantiations -- what any other quadruped would run -- on the
    same problem: both must match the oracle, and each other, on a chain with lift, impulse and aux stages."""
    monkeypatch.setenv("IDOCP_GENERAL_AXES", "1")
    m, o, g_general, q, v, h = make_pair(30, 1.55, 2, referee=True)
    monkeypatch.delenv("IDOCP_GENERAL_AXES")
    _, _, g_special, _, _ = make_pair(30, 1.55, 2)
    qq = q.copy()
    qq[7:] += 0.02 * np.random.default_rng(4).uniform(-1, 1, 12)
    assert o.update(0.0, qq, v) == 0 and g_general.update(0.0, qq, v) == 0 and g_special.update(0.0, qq, v) == 0 and h.update(0.0, qq, v) == 0
    M = len(o.chain(0.0))
    dirs = list(OCP_DIR_FIELDS) + ["dxi"]
    # (this grid has a 1.7 ms stage in front of the second impulse: stage by stage the FP64 oracle is itself a few 1e-10 away from its
    #  long double build there, so the referee decides; 1e-8 is the cap against the oracle and between the two instantiations)
    for f in dirs:
        hf = h.get_chain(f, M)
        referee_check(g_general.get_chain(f, M), o.get_chain(f, M), hf, f + " (general axes)")
        referee_check(g_special.get_chain(f, M), o.get_chain(f, M), hf, f + " (compile-time axes)")
    compare_chain(o, g_general, M, dirs, 1e-8, "direction (general axes)")
    compare_chain(o, g_special, M, dirs, 1e-8, "direction (compile-time axes)")
    # the two instantiations against each other: 2e-10 on every stage where the oracle sits on its referee (most of the chain), the
    # referee-derived bound on the stages around the short one -- no flat 1e-8 that a regression of one instantiation could hide behind
    exact_stages = M
    for f in dirs:
        _, n_exact = pairwise_check(g_general.get_chain(f, M), g_special.get_chain(f, M), o.get_chain(f, M), h.get_chain(f, M), f)
        exact_stages = min(exact_stages, n_exact)
    assert exact_stages >= M // 2, exact_stages


class _SequenceRecorder:
    """Collects what trotting_sequence pushes: [(active, points, time)], the phase list an MPC loop draws from."""
    def __init__(self):
        self.ev = []

    def set_contact_status(self, a, p):
        self.ev.append((list(a), np.array(p).copy(), None))

    def push_back_contact_status(self, a, p, t):
        self.ev.append((list(a), np.array(p).copy(), t))


def test_receding_horizon_mpc_loop_with_pop_front_and_push_back():
    """the receding-horizon use of OCPSolver -- popFrontContactStatus when the first discrete event has passed,
    pushBackContactStatus of the gait's next phase when it enters the horizon (ocp_solver.cpp:174-194 -> ContactSequence::pop_front /
    push_back, contact_sequence.hxx:63-160), the initial time advancing by two grid steps per MPC step, the measured state = the
    plan's own state at the new time.  GPU, FP64 oracle and the long double referee walk the same loop; after EVERY shift the chain
    (kind, index, slot, dt, contact dimension, switching rows) equals the oracle's discretiser and the Newton direction passes
    parity() (1e-10 against the oracle, else the referee rule), cap 1e-7.
    Two properties of the reference shape the loop: (i) initConstraints covers the event stages of the discretisation it is called
    on only (ocp_linearizer.cpp:40-71) and ConstraintComponentData starts as slack = dual = 0 (constraint_component_data.hxx:11-17),
    so an event index entering the horizon for the first time has no interior-point state: the loop calls initConstraints(t) after
    every pushBack, as a driver of the reference has to; (ii) the stage records do not move when an event is popped (impulse index
    k afterwards holds what index k held before: hybrid_container.hpp) -- stale warm starts, the same ones on both sides.
    Event and stage times stay clear of the grid and of the gait reference's phase boundaries (t0 = 0.013, events at 0.52 + 0.5 k),
    where floor() in the discretiser / the trotting reference would decide differently in FP64 and in long double."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    N, T, E = 31, 1.55, 4
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)
    g = HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=E)
    solvers = (o, h, g)
    rec = _SequenceRecorder()
    trotting_sequence(rec, m, 12, t_start=0.52)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        s.set_contact_status(rec.ev[0][0], rec.ev[0][1])
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    t0, dt_mpc = 0.013, 0.1
    state = dict(k=1, times=[])

    def feed(t):                       # push the phases whose switching time has entered the horizon
        n = 0
        while rec.ev[state["k"]][2] < t + T - 0.05:
            for s in solvers:
                s.push_back_contact_status(*rec.ev[state["k"]])
            state["times"].append(rec.ev[state["k"]][2])
            state["k"] += 1
            n += 1
        return n

    def same_chain(t):
        co, cg = o.chain(t), g.chain(t)
        assert len(co) == len(cg)
        for a, b in zip(co, cg):
            assert a["kind"] == b["kind"] and a["index"] == b["index"] and a["slot"] == b["slot"] and a["dimf"] == b["dimf"], (t, a, b)
            assert abs(a["dt"] - b["dt"]) < 1e-14 and (a["sw_event"] >= 0) == (b["sw_dimi"] > 0)
        return co

    t = t0
    feed(t)
    for s in solvers:
        s.init_constraints(t)
    for it in range(20):
        for s in solvers:
            assert s.update(t, q, v) == 0
    assert o.kkt_error(t, q, v) < 1e-8 and g.kkt_error(t, q, v)[0] < 1e-8
    pops, shapes, worst = 0, set(), 0.0
    for step in range(22):
        co = o.chain(t)
        tn = t0 + dt_mpc * (step + 1)
        at = [p for p, c in enumerate(co) if c["kind"] in ("stage", "terminal") and abs(c["t"] - tn) < 1e-9]
        assert at
        q, v = o.get_chain("q", len(co))[at[0]].copy(), o.get_chain("v", len(co))[at[0]].copy()       # the plan's state at the new time
        t = tn
        while state["times"] and state["times"][0] <= t + 1e-9:
            for s in solvers:
                s.pop_front_contact_status()
            state["times"].pop(0)
            pops += 1
        if feed(t):
            for s in solvers:
                s.init_constraints(t)
        co = same_chain(t)
        M = len(co)
        shapes.add("".join(c["kind"][0] for c in co))
        for sweep in range(3):
            for s in solvers:
                assert s.update(t, q, v) == 0
            for f in list(OCP_DIR_FIELDS) + ["dxi"]:
                worst = max(worst, parity(g.get_chain(f, M), o.get_chain(f, M), lambda: h.get_chain(f, M), (step, sweep, f), cap=1e-7))
        for f in ("q", "v", "a", "u", "f", "lmd", "gmm"):
            parity(g.get_chain(f, M), o.get_chain(f, M), lambda: h.get_chain(f, M), (step, "iterate", f), cap=1e-7)
        e_o, e_g = o.kkt_error(t, q, v), g.kkt_error(t, q, v)
        assert abs(e_g[0] - e_o) <= 1e-7 * max(1.0, e_o) and abs(e_g[1] - e_g[0]) <= 1e-9 * max(1.0, e_o)
    assert pops >= 4 and len(shapes) >= 5          # a lift and three impulse events left through the front; the chain changed shape
    print("MPC loop: %d pops, %d chain shapes, worst GPU-oracle direction distance %.2e" % (pops, len(shapes), worst))


def perturbed_states(m, n, seed=20250, dq=0.02, dv=0.05, base=ANYMAL_Q_STANDING):
    """Per-instance start states in the manner of SURVEY 8(d) C3 (base xy and joints +- 0.02 from std::mt19937_64(seed + b) there; numpy's
    generator here), plus what C3 leaves at zero: a base yaw / pitch of a few degrees and a non-zero velocity of every degree of freedom."""
    qs, vs = [], []
    for b in range(n):
        rng = np.random.default_rng(seed + b)
        q = base.copy()
        q[0:2] += dq * rng.uniform(-1, 1, 2)
        quat = np.array([0.0, 0.0, 0.0, 1.0]) + 0.03 * rng.normal(size=4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] += dq * rng.uniform(-1, 1, 12)
        qs.append(q)
        vs.append(dv * rng.uniform(-1, 1, m.nv))
    return np.array(qs), np.array(vs)


def test_full_size_c3_from_three_perturbed_states():
    """BASELINE configs[2] at its own size from per-instance states with a tilted base and v != 0 (the other full-size tests start every
    instance from the standing pose at rest): one GPU handle with three different instances against three oracle / referee pairs, first
    direction along the 120-stage chain under parity() with the 2e-8 cap of the standing-start test, then the iterates after four steps."""
    nimp = 9
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    T, N, B = 0.5 + nimp * 0.5 + 0.05, 100, 3
    qs, vs = perturbed_states(m, B)
    g = HipOCP(m, cost, cons, T, N, batch=B, max_num_impulse=nimp + 1)
    trotting_sequence(g, m, nimp)
    g.set_solution_batch("q", qs)
    g.set_solution_batch("v", vs)
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    pairs = []
    for b in range(B):
        o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
        h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)
        for s in (o, h):
            trotting_sequence(s, m, nimp)
            s.set_solution("q", qs[b])
            s.set_solution("v", vs[b])
            s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
            s.init_constraints(0.0)
        pairs.append((o, h))
    M = len(pairs[0][0].chain(0.0))
    assert M == 120
    e_g = g.kkt_error(0.0, qs, vs)
    assert g.update(0.0, qs, vs) == 0
    worst = 0.0
    for b, (o, h) in enumerate(pairs):
        e_o = o.kkt_error(0.0, qs[b], vs[b])
        assert abs(e_g[b] - e_o) <= 1e-9 * max(1.0, e_o)
        assert o.update(0.0, qs[b], vs[b]) == 0 and h.update(0.0, qs[b], vs[b]) == 0
        for f in list(OCP_DIR_FIELDS) + ["dxi"]:
            worst = max(worst, parity(g.get_chain(f, M, b), o.get_chain(f, M), lambda: h.get_chain(f, M), (b, f), cap=2e-8))
        ao, bo = o.step_sizes()
        ag, bg = g.step_sizes()
        assert abs(ag[b] - ao) < 1e-9 and abs(bg[b] - bo) < 1e-9
    assert np.abs(g.get_chain("dq", M, 0) - g.get_chain("dq", M, 1)).max() > 1e-3          # the instances are different problems
    for it in range(4):
        assert g.update(0.0, qs, vs) == 0
        for b, (o, h) in enumerate(pairs):
            assert o.update(0.0, qs[b], vs[b]) == 0 and h.update(0.0, qs[b], vs[b]) == 0
    for b, (o, h) in enumerate(pairs):
        for f in ("q", "v", "a", "u", "f"):
            parity(g.get_chain(f, M, b), o.get_chain(f, M), lambda: h.get_chain(f, M), (b, "iterate", f), cap=1e-7)
    print("configs[2] from three perturbed states: worst GPU-oracle distance of the first direction %.2e" % worst)


def test_configs4_grid_from_three_perturbed_states():
    """BASELINE configs[4] on its own grid (running gait, N = 200, T = 7 * 200 / 240: 26 touch-downs, 14 lift-offs, flight phases) from
    three per-instance states with a tilted base and v != 0: first direction along the chain under parity(), 1e-8 cap (the cap of the
    standing-start test of this gait)."""
    from helpers import ANYMAL_Q_RUNNING_START, running_problem, running_sequence
    m = anymal_model()
    steps = 10
    cost, cons = running_problem(m, steps)
    N, T, E, B = 200, 7.0 * 200 / 240, (steps + 3) * 2, 3
    qs, vs = perturbed_states(m, B, seed=20260, base=ANYMAL_Q_RUNNING_START)
    g = HipOCP(m, cost, cons, T, N, batch=B, max_num_impulse=E)
    assert running_sequence(g, m, steps) == 40
    g.set_solution_batch("q", qs)
    g.set_solution_batch("v", vs)
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    assert g.update(0.0, qs, vs) == 0
    worst = 0.0
    for b in range(B):
        o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
        h = OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)
        for s in (o, h):
            assert running_sequence(s, m, steps) == 40
            s.set_solution("q", qs[b])
            s.set_solution("v", vs[b])
            s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
            s.init_constraints(0.0)
            assert s.update(0.0, qs[b], vs[b]) == 0
        M = len(o.chain(0.0))
        for f in list(OCP_DIR_FIELDS) + ["dxi"]:
            worst = max(worst, parity(g.get_chain(f, M, b), o.get_chain(f, M), lambda: h.get_chain(f, M), (b, f), cap=1e-8))
    print("configs[4] grid from three perturbed states: worst GPU-oracle distance of the first direction %.2e" % worst)


def test_long_receding_horizon_run_recycles_the_event_slots():
    """The loop above for NINE seconds of gait instead of two: 90 shifts, every impulse / aux / lift slot of the containers (max_num_impulse = 4) reused
    several times over as events enter at the back and leave through the front.  GPU against the FP64 oracle at every tick: the same chain, the same
    first direction of the tick (cap 1e-6: the loop feeds both with the ORACLE's plan, so rounding differences of the iterates accumulate in the GPU's
    warm start only through its own multipliers), a KKT error that stays where the oracle's stays, no error code on the way."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    N, T, E = 31, 1.55, 4
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    g = HipOCP(m, cost, cons, T, N, batch=1, max_num_impulse=E)
    solvers = (o, g)
    rec = _SequenceRecorder()
    trotting_sequence(rec, m, 44, t_start=0.52)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        s.set_contact_status(rec.ev[0][0], rec.ev[0][1])
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    t0, dt_mpc = 0.013, 0.1
    k, times = 1, []

    def feed(t):
        nonlocal k
        n = 0
        while k < len(rec.ev) and rec.ev[k][2] < t + T - 0.05:
            for s in solvers:
                s.push_back_contact_status(*rec.ev[k])
            times.append(rec.ev[k][2])
            k += 1
            n += 1
        return n

    t = t0
    feed(t)
    for s in solvers:
        s.init_constraints(t)
    for it in range(20):
        for s in solvers:
            assert s.update(t, q, v) == 0
    pops, slots_used, worst, worst_kkt = 0, {}, 0.0, 0.0
    for step in range(90):
        co = o.chain(t)
        tn = t0 + dt_mpc * (step + 1)
        at = [p for p, c in enumerate(co) if c["kind"] in ("stage", "terminal") and abs(c["t"] - tn) < 1e-9]
        assert at, step
        q, v = o.get_chain("q", len(co))[at[0]].copy(), o.get_chain("v", len(co))[at[0]].copy()
        t = tn
        while times and times[0] <= t + 1e-9:
            for s in solvers:
                s.pop_front_contact_status()
            times.pop(0)
            pops += 1
        if feed(t):
            for s in solvers:
                s.init_constraints(t)
        co, cg = o.chain(t), g.chain(t)
        assert len(co) == len(cg), step
        for a, b in zip(co, cg):
            assert a["kind"] == b["kind"] and a["index"] == b["index"] and a["slot"] == b["slot"] and a["dimf"] == b["dimf"] and abs(a["dt"] - b["dt"]) < 1e-14, (step, a, b)
            if a["kind"] in ("impulse", "aux", "lift"):
                slots_used[(a["kind"], a["index"])] = slots_used.get((a["kind"], a["index"]), 0) + 1
        M = len(co)
        for sweep in range(2):
            assert o.update(t, q, v) == 0 and g.update(t, q, v) == 0, (step, sweep, capi.lib().idocp_last_error())
            if sweep == 0:
                for f in ("dq", "dv", "du", "df", "dlmd", "dgmm"):
                    e = rel_err(g.get_chain(f, M), o.get_chain(f, M))
                    worst = max(worst, e)
                    assert e < 1e-6, (step, f, e)
        e_o, e_g = o.kkt_error(t, q, v), g.kkt_error(t, q, v)[0]
        worst_kkt = max(worst_kkt, e_g)
        assert np.isfinite(e_g) and abs(e_g - e_o) <= 1e-5 * max(1.0, e_o), (step, e_g, e_o)
    assert pops >= 16 and max(slots_used.values()) >= 20 and len(slots_used) >= 6, (pops, slots_used)
    print("long MPC run: %d shifts, %d pops, slot use %s, worst direction distance %.2e, largest KKT error %.2e" % (90, pops, dict(sorted(slots_used.items())), worst, worst_kkt))
