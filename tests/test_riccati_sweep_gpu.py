"""The two forms of the backward Riccati sweep S3 (riccati_recursion_solver.cpp:48-107, backward_riccati_recursion_factorizer.hxx:44-161,
split_riccati_factorizer.hxx:36-101): one wavefront per instance with P in registers (ocp_riccati_backward_reg_kernel: batches of instances,
the kernel of the headline metric) and eight wavefronts per instance with P staged in LDS (ocp_riccati_backward_kernel<512>: a handful of
instances, the latency form).  A handle picks one by its batch size (idocp_ocp_riccati_sweep); the parity tests of the other files run small
batches, i.e. the latency form, so this file (i) compares the two forms with each other, stage by stage, and (ii) re-runs the oracle parity
tests of the other files with every OCPSolver handle FORCED into the register-resident form (idocp_ocp_set_riccati_sweep through
helpers.force_forms)."""
import ctypes as C

import numpy as np
import pytest

from helpers import ANYMAL_Q_STANDING, force_forms, OCP_DIR_FIELDS, HipOCP, anymal_model, anymal_problem, rel_err, trotting_sequence
from idocp_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture
def narrow(monkeypatch):
    force_forms(monkeypatch, sweep=0)


def _trotting_handle(mode, N, T, nimp, batch, monkeypatch):
    force_forms(monkeypatch, sweep=mode)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    lib = capi.lib()
    lib.idocp_ocp_riccati_sweep.argtypes = [C.c_void_p]
    assert lib.idocp_ocp_riccati_sweep(g.h) == mode
    trotting_sequence(g, m, nimp)
    g.set_solution("q", ANYMAL_Q_STANDING)
    g.set_solution("v", np.zeros(m.nv))
    g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init_constraints(0.0)
    return m, g


@pytest.mark.parametrize("N,T,nimp", [(31, 1.55, 2), (100, 5.05, 9)])
def test_the_two_forms_agree_stage_by_stage(N, T, nimp, monkeypatch):
    """Same problem, same iterates, two kernels with their own summation orders (tile products on 16x16x4 / 4x4x4 matrix instructions in
    different blockings): the directions agree to rounding times the conditioning of the chain, the iterates over six SQP steps to 1e-7."""
    m, a = _trotting_handle(0, N, T, nimp, 3, monkeypatch)
    _, b = _trotting_handle(1, N, T, nimp, 3, monkeypatch)
    rng = np.random.default_rng(11)
    qs = np.tile(ANYMAL_Q_STANDING, (3, 1))
    qs[:, 7:] += 0.02 * rng.uniform(-1, 1, (3, 12))
    vs = 0.05 * rng.uniform(-1, 1, (3, m.nv))
    M = len(a.chain(0.0))
    worst = 0.0
    for it in range(6):
        assert a.update(0.0, qs, vs) == 0 and b.update(0.0, qs, vs) == 0
        # (N = 100: the chain with the ill-conditioned stage of configs[2], where two FP64 evaluation orders separate by 5e-9: DESIGN 6b)
        tol = (1e-10 if N < 100 else 2e-8) if it == 0 else 1e-7
        for inst in range(3):
            for f in list(OCP_DIR_FIELDS) + ["dxi"]:
                e = rel_err(b.get_chain(f, M, inst), a.get_chain(f, M, inst))
                if it == 0:
                    worst = max(worst, e)
                assert e < tol, (it, inst, f, e)
    print("two sweep forms, first direction, worst relative difference: %.2e" % worst)
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu"):
        assert rel_err(b.get_chain(f, M, 1), a.get_chain(f, M, 1)) < 1e-7, f


def test_oracle_parity_of_the_register_resident_form_event_free(narrow):
    import test_ocp_gpu as T
    T.test_first_iteration_direction_parity(20, 1.0, 0.0)
    T.test_multi_iteration_parity_and_convergence()
    T.test_batch_instances_and_ragged_sizes()
    T.test_state_feedback_gain()


@pytest.mark.parametrize("N,T,tol", [(31, 1.55, 1e-10), (30, 1.55, None)])
def test_oracle_parity_of_the_register_resident_form_on_the_trotting_chain(narrow, N, T, tol):
    import test_hybrid_gpu as H
    H.test_first_iteration_direction_parity_along_the_chain(False, N, T, tol)


def test_oracle_parity_of_the_register_resident_form_flight_odd_contacts_and_mpc_loop(narrow):
    import test_hybrid_gpu as H
    H.test_flight_phase_sequence_parity()
    H.test_odd_contact_counts_take_the_general_class()
    H.test_hybrid_convergence_and_kkt_error_parity()
    H.test_receding_horizon_mpc_loop_with_pop_front_and_push_back()


def test_oracle_parity_of_the_register_resident_form_full_size(narrow):
    import test_hybrid_gpu as H
    H.test_full_size_c3_from_three_perturbed_states()
    H.test_configs4_grid_from_three_perturbed_states()


def test_oracle_parity_of_the_register_resident_form_plugins_and_line_search(narrow):
    import test_contact_distance_gpu as CD
    import test_joint_acceleration_limits_gpu as JA
    import test_ocp_gpu as T
    for mod in (CD, JA):
        for name in dir(mod):
            if name.startswith("test_") and "parnmpc" not in name:
                fn = getattr(mod, name)
                if callable(fn) and fn.__code__.co_argcount == 0:
                    fn()
    JA.test_ocp_uniform_horizon(True, True)
    T.test_line_search_accepted_steps_follow_the_oracle()


def test_default_choice_by_batch_size():
    """A handful of instances take the latency form, batches the register-resident sweep; the switch is per handle and survives a clone."""
    m = anymal_model()
    cost, cons = anymal_problem(m)
    lib = capi.lib()
    lib.idocp_ocp_riccati_sweep.argtypes = [C.c_void_p]
    lib.idocp_ocp_set_riccati_sweep.argtypes = [C.c_void_p, C.c_int]
    small, big = HipOCP(m, cost, cons, 0.5, 10, batch=4), HipOCP(m, cost, cons, 0.5, 10, batch=512)
    assert lib.idocp_ocp_riccati_sweep(small.h) == 1 and lib.idocp_ocp_riccati_sweep(big.h) == 0
    assert lib.idocp_ocp_set_riccati_sweep(small.h, 0) == 0 and lib.idocp_ocp_riccati_sweep(small.h) == 0
    assert lib.idocp_ocp_set_riccati_sweep(small.h, 2) != 0
    c = C.c_void_p()
    lib.idocp_ocp_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.idocp_ocp_clone(small.h, C.byref(c)) == 0
    assert lib.idocp_ocp_riccati_sweep(c) == 0
    lib.idocp_ocp_destroy(c)


def test_a_captured_graph_follows_the_switch(monkeypatch):
    """idocp_ocp_update_solution_graph replays captured launches: switching the form of the sweep must invalidate the capture.  A handle
    captures its iteration in the eight-wavefront form; a clone of it (same iterate, same form) and the handle itself are then switched to
    the register-resident form; the clone steps eagerly, the handle through the graph: bitwise the same iterate (the two forms differ in
    the last digits, so a stale capture shows)."""
    import copy
    torch = pytest.importorskip("torch")
    m, r = _trotting_handle(1, 31, 1.55, 2, 2, monkeypatch)
    lib = capi.lib()
    lib.idocp_ocp_set_riccati_sweep.argtypes = [C.c_void_p, C.c_int]
    lib.idocp_ocp_riccati_sweep.argtypes = [C.c_void_p]
    lib.idocp_ocp_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    q = np.tile(ANYMAL_Q_STANDING, (2, 1))
    q[:, 7:] += 0.01
    dq = torch.tensor(q, dtype=torch.float64, device="cuda")
    dv = torch.zeros((2, m.nv), dtype=torch.float64, device="cuda")
    args = (C.c_double(0.0), C.c_void_p(dq.data_ptr()), C.c_void_p(dv.data_ptr()))
    for _ in range(3):      # the first call launches eagerly, the second captures, the third replays -- in the eight-wavefront form
        capi.check(lib.idocp_ocp_update_solution_graph(r.h, *args), "graph")
    capi.check(lib.idocp_ocp_synchronize(r.h))
    h2 = C.c_void_p()
    capi.check(lib.idocp_ocp_clone(r.h, C.byref(h2)), "clone")
    e = copy.copy(r)
    e.h = h2
    assert lib.idocp_ocp_riccati_sweep(e.h) == 1
    for s_ in (r, e):
        assert lib.idocp_ocp_set_riccati_sweep(s_.h, 0) == 0 and lib.idocp_ocp_riccati_sweep(s_.h) == 0
    for _ in range(2):
        capi.check(lib.idocp_ocp_update_solution_graph(r.h, *args), "graph after the switch")
        capi.check(lib.idocp_ocp_update_solution_device(e.h, *args), "eager")
    capi.check(lib.idocp_ocp_synchronize(r.h))
    capi.check(lib.idocp_ocp_synchronize(e.h))
    M = len(r.chain(0.0))
    for f in ("q", "v", "a", "u", "f", "lmd", "gmm"):
        assert np.array_equal(e.get_chain(f, M), r.get_chain(f, M)), f
    # and the eight-wavefront form does differ in the last digits on this problem: the comparison above can tell a stale capture
    w = copy.copy(r)
    h3 = C.c_void_p()
    capi.check(lib.idocp_ocp_clone(r.h, C.byref(h3)), "clone")
    w.h = h3
    assert lib.idocp_ocp_set_riccati_sweep(w.h, 1) == 0
    capi.check(lib.idocp_ocp_update_solution_device(w.h, *args), "eager, eight wavefronts")
    capi.check(lib.idocp_ocp_update_solution_device(e.h, *args), "eager, one wavefront")
    capi.check(lib.idocp_ocp_synchronize(w.h))
    capi.check(lib.idocp_ocp_synchronize(e.h))
    assert any(not np.array_equal(e.get_chain(f, M), w.get_chain(f, M)) for f in ("q", "v", "a", "u", "f", "lmd", "gmm"))
