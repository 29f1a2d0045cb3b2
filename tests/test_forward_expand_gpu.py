"""The forward Riccati sweep that expands as it walks (round 5: ocp_forward_expand_kernel = S4 + K6 + the step-size reduction of
riccati_recursion_solver.cpp:110-251 in one kernel, one wavefront per instance) against the three kernels it replaces and against the
oracle.  A handle picks the fused form by its batch size (idocp_ocp_fused_forward; the parity tests elsewhere run small batches, i.e.
S4 + K6), so this file (i) compares the two forms with each other on one problem, stage by stage, and (ii) re-runs the oracle parity
tests of the other files with every OCPSolver handle FORCED into the fused form (idocp_ocp_set_fused_forward through
helpers.force_forms): event-free horizons, trotting / flight / odd-contact chains with impulse, aux and lift
stages and switching constraints, the full-size configs, plug-ins whose IPM rows live in the ext record, the filter line search."""
import ctypes as C

import numpy as np
import pytest

from helpers import ANYMAL_Q_STANDING, force_forms, OCP_DIR_FIELDS, HipOCP, anymal_model, anymal_problem, rel_err, trotting_sequence
from idocp_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture
def fused(monkeypatch):
    force_forms(monkeypatch, fused=1)


def _trotting_handle(mode, N, T, nimp, batch, monkeypatch):
    force_forms(monkeypatch, fused=mode)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    lib = capi.lib()
    lib.idocp_ocp_fused_forward.argtypes = [C.c_void_p]
    assert lib.idocp_ocp_fused_forward(g.h) == mode
    trotting_sequence(g, m, nimp)
    g.set_solution("q", ANYMAL_Q_STANDING)
    g.set_solution("v", np.zeros(m.nv))
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    return m, g


@pytest.mark.parametrize("N,T,nimp", [(31, 1.55, 2), (100, 5.05, 9)])
def test_fused_forward_equals_s4_k6_stage_by_stage(N, T, nimp, monkeypatch):
    """Same problem, same iterates: the fused walk forms dv+ as dv + dt (w - t) + Fv instead of Fvq dq + Fvv dv + Fvu du + Fv
    (contact_dynamics.hxx:150-156 substituted) and sums P dx, K dx in the order of K6 / S4, so the two forms agree to rounding along the
    whole chain -- direction, step sizes, and the iterates over six SQP steps."""
    m, a = _trotting_handle(0, N, T, nimp, 3, monkeypatch)
    _, b = _trotting_handle(1, N, T, nimp, 3, monkeypatch)
    rng = np.random.default_rng(7)
    qs = np.tile(ANYMAL_Q_STANDING, (3, 1))
    qs[:, 7:] += 0.02 * rng.uniform(-1, 1, (3, 12))
    vs = 0.05 * rng.uniform(-1, 1, (3, m.nv))
    M = len(a.chain(0.0))
    for it in range(6):
        assert a.update(0.0, qs, vs) == 0 and b.update(0.0, qs, vs) == 0
        # (N = 100: the chain with the ill-conditioned stage of configs[2], where two FP64 evaluation orders separate by 5e-9: DESIGN 6b)
        tol = (1e-11 if N < 100 else 2e-8) if it == 0 else 1e-7
        for inst in range(3):
            for f in list(OCP_DIR_FIELDS) + ["dxi"]:
                e = rel_err(b.get_chain(f, M, inst), a.get_chain(f, M, inst))
                assert e < tol, (it, inst, f, e)
        pa, da = a.step_sizes()
        pb, db = b.step_sizes()
        assert np.abs(pa - pb).max() < 1e-9 and np.abs(da - db).max() < 1e-9
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu"):
        assert rel_err(b.get_chain(f, M, 1), a.get_chain(f, M, 1)) < 1e-7, f


def test_oracle_parity_of_the_fused_form_event_free(fused):
    import test_ocp_gpu as T
    T.test_first_iteration_direction_parity(20, 1.0, 0.0)
    T.test_multi_iteration_parity_and_convergence()
    T.test_batch_instances_and_ragged_sizes()
    T.test_state_feedback_gain()


@pytest.mark.parametrize("N,T,tol", [(31, 1.55, 1e-10), (30, 1.55, None)])
def test_oracle_parity_of_the_fused_form_on_the_trotting_chain(fused, N, T, tol):
    import test_hybrid_gpu as H
    H.test_first_iteration_direction_parity_along_the_chain(False, N, T, tol)


def test_oracle_parity_of_the_fused_form_flight_odd_contacts_and_mpc_loop(fused):
    import test_hybrid_gpu as H
    H.test_flight_phase_sequence_parity()
    H.test_odd_contact_counts_take_the_general_class()
    H.test_hybrid_convergence_and_kkt_error_parity()
    H.test_receding_horizon_mpc_loop_with_pop_front_and_push_back()


def test_oracle_parity_of_the_fused_form_full_size(fused):
    import test_hybrid_gpu as H
    H.test_full_size_c3_from_three_perturbed_states()
    H.test_configs4_grid_from_three_perturbed_states()


def test_oracle_parity_of_the_fused_form_plugins_and_line_search(fused):
    import test_contact_distance_gpu as CD
    import test_joint_acceleration_limits_gpu as JA
    import test_ocp_gpu as T
    for name in dir(CD):
        if name.startswith("test_") and "parnmpc" not in name:
            fn = getattr(CD, name)
            if callable(fn) and fn.__code__.co_argcount == 0:
                fn()
    for name in dir(JA):
        if name.startswith("test_") and "parnmpc" not in name:
            fn = getattr(JA, name)
            if callable(fn) and fn.__code__.co_argcount == 0:
                fn()
    JA.test_ocp_uniform_horizon(True, True)
    T.test_line_search_accepted_steps_follow_the_oracle()


def test_default_choice_by_batch_size():
    """Small batches keep S4 + K6 (latency mode: K6 spreads the stages of a few instances over the chip), batches of instances take the
    fused walk; the switch is per handle and survives a clone."""
    m = anymal_model()
    cost, cons = anymal_problem(m)
    lib = capi.lib()
    lib.idocp_ocp_fused_forward.argtypes = [C.c_void_p]
    lib.idocp_ocp_set_fused_forward.argtypes = [C.c_void_p, C.c_int]
    small, big = HipOCP(m, cost, cons, 0.5, 10, batch=4), HipOCP(m, cost, cons, 0.5, 10, batch=512)
    assert lib.idocp_ocp_fused_forward(small.h) == 0 and lib.idocp_ocp_fused_forward(big.h) == 1
    assert lib.idocp_ocp_set_fused_forward(small.h, 1) == 0 and lib.idocp_ocp_fused_forward(small.h) == 1
    assert lib.idocp_ocp_set_fused_forward(small.h, 2) != 0
    c = C.c_void_p()
    lib.idocp_ocp_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.idocp_ocp_clone(small.h, C.byref(c)) == 0
    assert lib.idocp_ocp_fused_forward(c) == 1
    lib.idocp_ocp_destroy(c)


def test_chains_longer_than_the_walk_keeps_in_lds_fall_back_to_s4_k6():
    """The fused walk keeps the chain (slot, status word, two time steps per node) in LDS: OcpForwardExpandMaxChain = 320 nodes.  A longer chain
    runs S4 + K6 whatever the handle was told, and is held to the oracle like any other."""
    from helpers import OracleOCP, anymal_contact_points
    m = anymal_model()
    cost, cons = anymal_problem(m)
    N, T = 330, 16.5
    g = HipOCP(m, cost, cons, T, N, batch=2)
    o = OracleOCP(m, cost, cons, T, N)
    lib = capi.lib()
    lib.idocp_ocp_fused_forward.argtypes = [C.c_void_p]
    lib.idocp_ocp_set_fused_forward.argtypes = [C.c_void_p, C.c_int]
    assert lib.idocp_ocp_set_fused_forward(g.h, 1) == 0
    pts = anymal_contact_points(m)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (g, o):
        s.set_contact_status([1, 1, 1, 1], pts)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    q[7:] += 0.02
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    assert lib.idocp_ocp_fused_forward(g.h) == 0               # 331 nodes: the chain does not fit
    for f in OCP_DIR_FIELDS:
        assert rel_err(g.get(f, 1), o.get(f)) < 1e-9, f


def test_side_stream_iteration_equals_the_single_stream_one_and_survives_graph_capture(monkeypatch):
    """Round 6: handles of >= 128 instances run the switching-constraint kernel, the impulse stages' nominal launch and the base-pose update
    on a side stream beside their independent neighbours (fork / join with events; ocp_capi.hip launchSwitchO / launchNominalO / joinSideO /
    launchIntegrateO).  (i) A handle created with IDOCP_SIDE_STREAM_MIN_BATCH above its batch (everything on one stream) and one with the
    side stream step the same trotting chain to BITWISE the same iterate -- the kernels and their inputs are the same, only the order in
    which independent launches start differs.  (ii) The fork / join is captured into the hipGraph of idocp_ocp_update_solution_graph:
    a clone stepping through the graph stays bitwise on the eager handle."""
    import copy
    torch = pytest.importorskip("torch")
    lib = capi.lib()
    lib.idocp_ocp_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    B = 128
    m = anymal_model()
    q = np.tile(ANYMAL_Q_STANDING, (B, 1))
    q[:, 7:] += 0.01 * np.random.default_rng(3).uniform(-1, 1, (B, 12))
    dq = torch.tensor(q, dtype=torch.float64, device="cuda")
    dv = torch.zeros((B, m.nv), dtype=torch.float64, device="cuda")
    args = (C.c_double(0.0), C.c_void_p(dq.data_ptr()), C.c_void_p(dv.data_ptr()))

    def handle():
        cost, cons = anymal_problem(m, trotting_ref=True)
        g = HipOCP(m, cost, cons, 1.55, 31, batch=B, max_num_impulse=3)
        trotting_sequence(g, m, 2)
        g.set_solution("q", ANYMAL_Q_STANDING)
        g.set_solution("v", np.zeros(m.nv))
        g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        g.init_constraints(0.0)
        return g

    side = handle()                                   # (the default: a side stream from batch 128 on)
    # the switch is read once per process: a handle WITHOUT a side stream is one below the threshold -- here made by a batch-127 twin would change
    # the problem, so the single-stream reference runs kernel by kernel through idocp_ocp_launch_kernel on a clone, ids in the plain order, with a
    # stream synchronisation after every launch (nothing can overlap)
    h2 = C.c_void_p()
    capi.check(lib.idocp_ocp_clone(side.h, C.byref(h2)), "clone")
    plain = copy.copy(side)
    plain.h = h2
    h3 = C.c_void_p()
    capi.check(lib.idocp_ocp_clone(side.h, C.byref(h3)), "clone")
    graph = copy.copy(side)
    graph.h = h3
    lib.idocp_ocp_fused_forward.argtypes = [C.c_void_p]
    kids = [0, 7, 8, 2, 3, 6] if lib.idocp_ocp_fused_forward(side.h) else [0, 7, 8, 2, 3, 4, 5, 6]
    for it in range(3):
        capi.check(lib.idocp_ocp_update_solution_device(side.h, *args), "eager, side stream")
        capi.check(lib.idocp_ocp_update_solution_graph(graph.h, *args), "graph")      # (first call eager, second captures, third replays)
        if it == 0:
            capi.check(lib.idocp_ocp_update_solution_device(plain.h, *args), "first iteration of the serialised clone: discretises and uploads")
        else:
            for kid in kids:
                capi.check(lib.idocp_ocp_launch_kernel(plain.h, kid, args[1], args[2]), "kernel %d" % kid)
                capi.check(lib.idocp_ocp_synchronize(plain.h))
    for s_ in (side, plain, graph):
        capi.check(lib.idocp_ocp_synchronize(s_.h))
    M = len(side.chain(0.0))
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu"):
        for inst in (0, B - 1):
            ref = side.get_chain(f, M, inst)
            assert np.isfinite(ref).all()
            assert np.array_equal(plain.get_chain(f, M, inst), ref), ("serialised", f, inst)
            assert np.array_equal(graph.get_chain(f, M, inst), ref), ("graph", f, inst)


def test_oracle_parity_with_the_side_stream_forced_on_small_batches(monkeypatch):
    """The parity tests of the other files run small batches, i.e. handles WITHOUT a side stream (it starts at 128 instances); here the chains with
    switching constraints, impulse stages and lift stages, the MPC loop and the full-size configs run with IDOCP_SIDE_STREAM_MIN_BATCH=1: every
    handle forks K5s / the impulse stages' nominal launch / the base poses onto its side stream and joins them, under the same oracle bars."""
    monkeypatch.setenv("IDOCP_SIDE_STREAM_MIN_BATCH", "1")
    import test_hybrid_gpu as H
    H.test_first_iteration_direction_parity_along_the_chain(False, 31, 1.55, 1e-10)
    H.test_flight_phase_sequence_parity()
    H.test_hybrid_convergence_and_kkt_error_parity()
    H.test_receding_horizon_mpc_loop_with_pop_front_and_push_back()
    H.test_full_size_c3_from_three_perturbed_states()
    import test_golden_kkt as K
    K.test_hip_direction_is_the_dense_newton_direction()
