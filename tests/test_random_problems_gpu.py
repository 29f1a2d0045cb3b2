"""Parity on RANDOM problems: horizon, time step, the feet in contact (none, one, two, three, all four -- every stage class of the condensation kernel,
the odd contact counts included), cost weights over four decades, the friction coefficient, the start state.  The hand-written cases of the other test
files sit at the corners someone thought of; this one draws from the space between them.  Bar: 1e-10 on the first Newton direction against the oracle,
stage by stage, or the long double referee's word where two FP64 evaluation orders cannot agree that far (helpers.parity)."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem,
                     parity)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def random_problem(rng, m):
    cost, cons = anymal_problem(m, trotting_ref=False)
    nv = m.nv
    lw = lambda lo, hi, n: 10.0 ** rng.uniform(lo, hi, size=n)
    cost.set("q_weight", lw(-1, 2, nv)).set("qf_weight", lw(-1, 2, nv)).set("v_weight", lw(-2, 1, nv)).set("vf_weight", lw(-2, 1, nv))
    cost.set("a_weight", lw(-4, -1, nv)).set("u_weight", lw(-5, -2, nv))
    for c in range(4):
        for k in range(3):
            cost.f_weight[c][k] = float(10.0 ** rng.uniform(-4, -2))
            cost.f_ref[c][k] = float(rng.uniform(-5, 5)) if k < 2 else float(rng.uniform(20, 120))
    cons.mu = float(rng.uniform(0.4, 1.0))
    return cost, cons


@pytest.mark.parametrize("seed", range(12))
def test_first_direction_on_random_problems(seed):
    rng = np.random.default_rng(1000 + seed)
    m = anymal_model()
    cost, cons = random_problem(rng, m)
    N = int(rng.integers(3, 25))
    T = N * float(rng.uniform(0.01, 0.05))
    active = [int(x) for x in rng.integers(0, 2, size=4)]
    if seed < 5:
        active = [[0, 0, 0, 0], [1, 0, 0, 0], [1, 0, 1, 0], [0, 1, 1, 1], [1, 1, 1, 1]][seed]      # every contact count at least once
    pts = anymal_contact_points(m)
    q = ANYMAL_Q_STANDING.copy()
    q[:3] += rng.uniform(-0.05, 0.05, 3)
    quat = q[3:7] + rng.uniform(-0.05, 0.05, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.uniform(-0.15, 0.15, 12)
    v = rng.uniform(-0.3, 0.3, m.nv)
    fz = [float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), float(rng.uniform(40, 110))]
    par = seed % 3 == 2                                       # every third problem through ParNMPCSolver
    Hip, Orc = (HipParNMPC, OracleParNMPC) if par else (HipOCP, OracleOCP)
    solvers = [Hip(m, cost, cons, T, N, batch=2), Orc(m, cost, cons, T, N), Orc(m, cost, cons, T, N, hp=True)]
    for s in solvers:
        s.set_contact_status(active, pts)
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", fz)
        s.init(0.0) if par else s.init_constraints(0.0)
    g, o, h = solvers
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    ran_referee = []

    def referee(name):
        if not ran_referee:
            assert h.update(0.0, q, v) == 0
            ran_referee.append(1)
        return h.get(name)

    worst = 0.0
    for name in OCP_DIR_FIELDS:
        worst = max(worst, parity(g.get(name, 1), o.get(name), lambda name=name: referee(name), (seed, active, N, name), tol=TOL, cap=1e-7))
    print("seed %d  contacts %s  N %d  %s  worst %.2e%s" % (seed, active, N, "ParNMPC" if par else "OCP", worst, "  (referee consulted)" if ran_referee else ""))


@pytest.mark.parametrize("seed", range(10))
def test_first_direction_along_random_event_chains(seed):
    """Random gaits: up to three events at random times, a random set of feet lifting off or touching down at each -- one, two, three or four feet at
    once, so the switching constraint has 3, 6, 9 or 12 rows (the trot of the other tests: always 6) and the impulse stages every row count.  First
    Newton direction along the whole chain (grid, impulse, aux and lift stages), every field, under helpers.parity."""
    from helpers import OCP_SOL_FIELDS
    rng = np.random.default_rng(500 + seed)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    N = int(rng.integers(12, 30))
    dt = float(rng.uniform(0.02, 0.04))
    T = N * dt
    E = 3
    solvers = [HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=E), OracleOCP(m, cost, cons, T, N, max_num_impulse=E),
               OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True)]
    active = np.array([1, 1, 1, 1]) if seed % 2 == 0 else rng.integers(0, 2, size=4)
    landing = seed == 0                                      # one case by hand: all four feet touch down at once out of a flight phase (12 rows)
    if landing:
        active = np.array([0, 0, 0, 0])
    for s in solvers:
        s.set_contact_status(active, pts)
    n_ev = int(rng.integers(1, E + 1))
    # events at least two grid intervals apart and off the grid points (the near-grid cases have tests of their own; here the rows are the subject)
    times = np.sort(rng.choice(np.arange(2, N - 2), size=n_ev, replace=False)) * dt + rng.uniform(0.2, 0.8, n_ev) * dt
    times = times[np.concatenate([[True], np.diff(times) > 2 * dt])]
    rows = []
    for t_ev in times:
        flip = np.zeros(4, dtype=int)
        flip[rng.choice(4, size=int(rng.integers(1, 5)), replace=False)] = 1
        if landing:
            flip[:] = 1
        nxt = np.where(flip == 1, 1 - active, active)
        rows.append(3 * int(((nxt == 1) & (active == 0)).sum()))
        for s in solvers:
            s.push_back_contact_status(nxt, pts, float(t_ev))
        active = nxt
    q = ANYMAL_Q_STANDING.copy()
    q[7:] += rng.uniform(-0.05, 0.05, 12)
    v = rng.uniform(-0.1, 0.1, m.nv)
    for s in solvers:
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    g, o, h = solvers
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    M = len(o.chain(0.0))
    assert len(g.chain(0.0)) == M
    ran = []

    def referee(name):
        if not ran:
            assert h.update(0.0, q, v) == 0
            ran.append(1)
        return h.get_chain(name, M)

    worst = 0.0
    for name in list(OCP_DIR_FIELDS) + ["dxi"]:
        worst = max(worst, parity(g.get_chain(name, M), o.get_chain(name, M), lambda name=name: referee(name), (seed, rows, name), tol=TOL, cap=1e-6))
    kinds = [c["kind"] for c in o.chain(0.0)]
    print("seed %d  N %d  touch-down rows per event %s  chain %d (%d impulse, %d lift)  worst %.2e%s" %
          (seed, N, rows, M, kinds.count("impulse"), kinds.count("lift"), worst, "  (referee consulted)" if ran else ""))


@pytest.mark.parametrize("seed", range(8))
def test_first_direction_along_random_event_chains_parnmpc(seed):
    """The same gaits through ParNMPCSolver (backward-Euler stages, the KKT inverses of the aux / impulse pair with 3, 6, 9 or 12 extra rows: the general
    instantiation of the event kernels).  The coarse update from a cold start is ill-conditioned around an event (tests/test_parnmpc_hybrid_gpu.py), so the
    long double referee decides stage by stage; the cap is on the distance from the FP64 oracle."""
    from helpers import HipParNMPC, OracleParNMPC
    rng = np.random.default_rng(900 + seed)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    N = int(rng.integers(12, 26))
    dt = float(rng.uniform(0.02, 0.04))
    T = N * dt
    E = 2
    solvers = [HipParNMPC(m, cost, cons, T, N, batch=2, max_num_impulse=E), OracleParNMPC(m, cost, cons, T, N, max_num_impulse=E),
               OracleParNMPC(m, cost, cons, T, N, max_num_impulse=E, hp=True)]
    active = np.array([0, 0, 0, 0]) if seed == 0 else (np.array([1, 1, 1, 1]) if seed % 2 == 0 else rng.integers(0, 2, size=4))
    for s in solvers:
        s.set_contact_status(active, pts)
    n_ev = int(rng.integers(1, E + 1))
    times = np.sort(rng.choice(np.arange(2, N - 3), size=n_ev, replace=False)) * dt + rng.uniform(0.2, 0.8, n_ev) * dt
    times = times[np.concatenate([[True], np.diff(times) > 2 * dt])]
    rows = []
    for t_ev in times:
        flip = np.zeros(4, dtype=int)
        flip[rng.choice(4, size=int(rng.integers(1, 5)), replace=False)] = 1
        if seed == 0:
            flip[:] = 1
        nxt = np.where(flip == 1, 1 - active, active)
        rows.append(3 * int(((nxt == 1) & (active == 0)).sum()))
        for s in solvers:
            s.push_back_contact_status(nxt, pts, float(t_ev))
        active = nxt
    q = ANYMAL_Q_STANDING.copy()
    q[7:] += rng.uniform(-0.03, 0.03, 12)
    v = rng.uniform(-0.05, 0.05, m.nv)
    for s in solvers:
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init(0.0)
    g, o, h = solvers
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    kinds = [c["kind"] for c in o.chain(0.0)]
    M = len(kinds)
    assert len(g.chain(0.0)) == M + 1
    ran = []

    def referee(name, keep):
        if not ran:
            assert h.update(0.0, q, v) == 0
            ran.append(1)
        return h.get_chain(name, M)[keep]

    keep_reg = np.array([k != "impulse" for k in kinds])
    worst = 0.0
    for name in OCP_DIR_FIELDS:
        keep = keep_reg if name in ("du", "dnu_passive") else np.ones(M, bool)
        worst = max(worst, parity(g.get_chain(name, M + 1)[:M][keep], o.get_chain(name, M)[keep], lambda name=name, keep=keep: referee(name, keep),
                                  (seed, rows, name), tol=TOL, cap=1e-5))
    print("seed %d  N %d  touch-down rows per event %s  chain %d (%d impulse, %d lift)  worst %.2e%s" %
          (seed, N, rows, M, kinds.count("impulse"), kinds.count("lift"), worst, "  (referee consulted)" if ran else ""))


@pytest.mark.parametrize("seed", range(6))
def test_filter_line_search_on_random_problems_follows_the_oracle(seed):
    """updateSolution(..., line_search = true) on random problems (OCPSolver: uniform contacts or a random event; the filter logic runs on the host,
    the trial iterates' cost and violation on the device): the accepted steps are the oracle's, iteration after iteration -- a different accept /
    reject decision anywhere would show as a different step size at once."""
    import ctypes as C
    from helpers import P, arr, rel_err
    from idocp_amd import capi
    rng = np.random.default_rng(4242 + seed)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False) if seed < 4 else random_problem(rng, m)      # (the last two: weights over four decades; an iterate
    pts = anymal_contact_points(m)                                                                  #  may then lose convexity -- both must say so together)
    N = int(rng.integers(8, 22))
    T = N * float(rng.uniform(0.02, 0.04))
    E = 1 if seed % 2 else 0
    g = HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=E)
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=E)
    o.lib.oracle_ocp_update_solution_ls.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
    active = np.array([1, 1, 1, 1])
    for s in (g, o):
        s.set_contact_status(active, pts)
    if E:
        nxt = np.array([0, 1, 1, 0]) if seed % 4 == 1 else np.array([1, 0, 1, 1])
        t_ev = (int(rng.integers(2, N - 2)) + float(rng.uniform(0.2, 0.8))) * T / N
        for s in (g, o):
            s.push_back_contact_status(nxt, pts, t_ev)
    q = ANYMAL_Q_STANDING.copy()
    q[7:] += rng.uniform(-0.1, 0.1, 12)
    v = rng.uniform(-0.2, 0.2, m.nv)
    for s in (g, o):
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    steps = []
    for it in range(5):
        ro = o.lib.oracle_ocp_update_solution_ls(o.h, 0.0, P(arr(q)), P(arr(v)))
        rg = g.lib.idocp_ocp_update_solution(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv)), 1)
        assert (ro == 0) == (rg == 0), (seed, it, "one of the two lost positive definiteness, the other did not", ro, rg, capi.lib().idocp_last_error())
        if ro != 0:
            steps.append(float("nan"))
            break
        (ao, bo), (ag, bg) = o.step_sizes(), g.step_sizes()
        assert abs(ag[1] - ao) <= 1e-9 * max(ao, 1e-3) and abs(bg[1] - bo) <= 1e-7 * max(bo, 1e-3), (seed, it, ag[1], ao, bg[1], bo)
        steps.append(ao)
        for f in ("q", "v", "a", "u", "f"):
            assert rel_err(g.get(f, 1), o.get(f)) < 1e-7, (seed, it, f)
    print("seed %d  N %d  event %d  accepted primal steps %s" % (seed, N, E, " ".join("%.3g" % a for a in steps)))


@pytest.mark.parametrize("seed", range(10))
def test_first_direction_on_random_fixed_base_problems(seed):
    """The fixed-base solvers on random problems: horizon, weights over four decades, tighter or looser joint limits, acceleration limits on or off, a
    task-space cost (3D / 6D, a random frame of the chain, a random reference pose, per-stage references or a constant one) on every other UnOCP
    problem; UnOCPSolver and UnParNMPCSolver alternate.  1e-10 on the first direction, the Riccati factors and the KKT error."""
    import ctypes as C
    from helpers import (DIR_FIELDS, HipUnOCP, HipUnParNMPC, OracleUnOCP, OracleUnParNMPC, iiwa14_model, rel_err)
    from idocp_amd import capi
    from idocp_amd.workloads import task_space_problem
    rng = np.random.default_rng(31337 + seed)
    m = iiwa14_model()
    nv = m.nv
    par = seed % 2 == 1
    task = (not par) and seed % 4 == 0
    lw = lambda lo, hi: 10.0 ** rng.uniform(lo, hi, size=nv)
    if task:
        dim = 3 if seed % 8 == 0 else 6
        cost, cons = task_space_problem(m, dim=dim, frame_id=int(rng.choice([10, 14, 18, 22])), weight=float(10.0 ** rng.uniform(1, 3)), time_varying=bool(seed % 8 == 4))
        ang = rng.uniform(-1, 1, 3)
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        Rm = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        ref = np.concatenate([Rm.ravel(), rng.uniform(-0.5, 0.8, 3)])
        for k in range(12):
            cost.task_ref[k] = float(ref[k])
    else:
        cost = capi.Cost()
        cons = capi.Constraints()
        capi.lib().idocp_constraints_init(C.byref(cons))
        cost.set("q_ref", rng.uniform(-1, 1, nv)).set("v_ref", rng.uniform(-1, 1, nv)).set("u_ref", rng.uniform(-5, 5, nv))
        cost.set("q_weight", lw(-1, 2)).set("qf_weight", lw(-1, 2))
    cost.set("v_weight", lw(-2, 1)).set("vf_weight", lw(-2, 1)).set("a_weight", lw(-3, -1)).set("u_weight", lw(-5, -2))
    for i in range(nv):
        m.u_max[i] = float(rng.uniform(40, 300))
        m.v_max[i] = float(rng.uniform(1.0, 3.0))
    if seed % 3 == 0:
        cons.joint_acceleration_lower_limit = cons.joint_acceleration_upper_limit = 1
        for i in range(nv):
            cons.a_min[i], cons.a_max[i] = float(-rng.uniform(5, 50)), float(rng.uniform(5, 50))
    N = int(rng.integers(2, 40))
    T = N * float(rng.uniform(0.01, 0.06))
    q, v = rng.uniform(-1.2, 1.2, nv), rng.uniform(-0.5, 0.5, nv)
    Hip, Orc = (HipUnParNMPC, OracleUnParNMPC) if par else (HipUnOCP, OracleUnOCP)
    g, o = Hip(m, cost, cons, T, N, batch=3), Orc(m, cost, cons, T, N)
    for s in (g, o):
        s.set_solution("q", 0.5 * q)
        s.set_solution("v", np.zeros(nv))
        if task and cost.task_time_varying:
            refs = np.tile(ref, (N + 1, 1))
            refs[:, 9:] += 0.05 * np.sin(np.arange(N + 1))[:, None]
            s.set_task_refs(refs)
        if par:
            s.init(0.0)
    eo, eg = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[2]
    assert abs(eg - eo) <= 1e-10 * max(1.0, eo)
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    worst = 0.0
    for f in DIR_FIELDS:
        a, b = (g.get(f, 2), o.get(f)) if par else (g.direction(f, 2), o.direction(f))
        e = rel_err(a, b)
        worst = max(worst, e)
        assert e <= TOL, (seed, f, e)
    if not par:
        for x, y in zip(g.riccati(2), o.riccati()):
            assert rel_err(x, y) <= TOL
    print("seed %d  %s  N %d  task %s  acceleration limits %s  worst %.2e" % (seed, "UnParNMPC" if par else "UnOCP", N, (dim if task else "-"), bool(seed % 3 == 0), worst))


@pytest.mark.parametrize("solver", ["ocp", "parnmpc"])
def test_a_ragged_batch_of_different_instances(solver, monkeypatch):
    """70 instances with states of their own along an event chain: 70 x 17 chain stages are neither a multiple of a wavefront's 64 lanes (the nominal sweeps carry
    one stage per lane) nor of anything else; both sweeps run in their throughput forms (one wavefront per instance, fused forward sweep -- the latter would start at
    batch 192 by itself and is switched on here).  Instances at the seams -- 0, 31, 63, 64, 69 -- against an oracle each, first direction along the whole chain."""
    from helpers import force_forms
    force_forms(monkeypatch, fused=1, sweep=0)
    rng = np.random.default_rng(7070)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    N, dt, B = 13, 0.03, 70
    par = solver == "parnmpc"
    Hip, Orc = (HipParNMPC, OracleParNMPC) if par else (HipOCP, OracleOCP)
    g = Hip(m, cost, cons, N * dt, N, batch=B, max_num_impulse=2)
    events = [([1, 0, 1, 1], 3.4 * dt), ([1, 1, 1, 1], 8.7 * dt)]
    qs = np.tile(ANYMAL_Q_STANDING, (B, 1))
    qs[:, 7:] += rng.uniform(-0.08, 0.08, (B, 12))
    qs[:, 0:3] += rng.uniform(-0.03, 0.03, (B, 3))
    quat = qs[:, 3:7] + rng.uniform(-0.03, 0.03, (B, 4))
    qs[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    vs = rng.uniform(-0.15, 0.15, (B, m.nv))
    fz = [0, 0, 0.25 * (-m.total_mass * m.gravity[2])]

    def prepare(s):
        s.set_contact_status([1, 1, 1, 1], pts)
        for status, when in events:
            s.push_back_contact_status(status, pts, when)
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", fz)
        s.init(0.0) if par else s.init_constraints(0.0)

    prepare(g)
    if not par:
        assert g.lib.idocp_ocp_riccati_sweep(g.h) == 0 and g.lib.idocp_ocp_fused_forward(g.h) == 1      # the throughput forms
    assert g.update(0.0, qs, vs) == 0
    worst = 0.0
    for b in (0, 31, 63, 64, 69):
        o, h = Orc(m, cost, cons, N * dt, N, max_num_impulse=2), Orc(m, cost, cons, N * dt, N, max_num_impulse=2, hp=True)
        prepare(o)
        prepare(h)
        assert o.update(0.0, qs[b], vs[b]) == 0
        M = len(o.chain(0.0))
        ran = []

        def referee(name, h=h, b=b, M=M, ran=ran):
            if not ran:
                assert h.update(0.0, qs[b], vs[b]) == 0
                ran.append(1)
            return h.get_chain(name, M)

        keep = np.array([c["kind"] != "impulse" for c in o.chain(0.0)])
        for name in OCP_DIR_FIELDS:
            k = keep if (par and name in ("du", "dnu_passive")) else np.ones(M, bool)
            have = g.get_chain(name, M + (1 if par else 0), b)[:M][k]
            worst = max(worst, parity(have, o.get_chain(name, M)[k], lambda name=name, k=k: referee(name)[k], (solver, b, name), tol=TOL, cap=1e-6))
    print("%s, 70 different instances: worst of five %.2e" % (solver, worst))
