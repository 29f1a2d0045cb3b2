"""Edges of the input space through the C ABI, on the GPU: the shortest horizons every solver accepts (N = 1, 2: one stage + the terminal one -- the
serial sweeps, the prefetch pipelines and the stage-class tables all have their corner there), non-finite numbers in the measured state or the cost, and
sizes no device can hold.  The bar: the same 1e-10 parity where the problem is well-posed; an error code and a message -- not a hang, not a crash, not
silent garbage -- where it is not; and the handle (or a fresh one) still works afterwards."""
import ctypes as C

import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, HipOCP, HipParNMPC, HipUnOCP, HipUnParNMPC, OracleOCP, OracleParNMPC, OracleUnOCP, OracleUnParNMPC, P,
                     anymal_contact_points, anymal_model, anymal_problem, arr, iiwa14_model, rel_err, unocp_problem)
from idocp_amd import capi

pytestmark = pytest.mark.gpu
TOL = 1e-10
UN_DIR = ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta")
OCP_DIR = ("dq", "dv", "da", "du", "df", "dlmd", "dgmm", "dbeta", "dmu", "dnu_passive")


@pytest.mark.parametrize("N", [1, 2, 3])
def test_shortest_horizons_fixed_base(N):
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    T = 0.05 * N
    q, v = np.full(m.nv, 0.7), np.full(m.nv, -0.2)
    for Hip, Orc in ((HipUnOCP, OracleUnOCP), (HipUnParNMPC, OracleUnParNMPC)):
        un = Hip is HipUnOCP
        g, o = Hip(m, cost, cons, T, N, batch=3), Orc(m, cost, cons, T, N)
        for s in (g, o):
            s.set_solution("q", q)
            s.set_solution("v", v)
            if not un:
                s.init(0.0)
        for it in range(2):
            assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
            for name in UN_DIR:
                a = g.direction(name, 2) if un else g.get(name, 2)
                b = o.direction(name) if un else o.get(name)
                assert a.shape == b.shape and rel_err(a, b) <= (TOL if it == 0 else 1e-9), (N, un, it, name, rel_err(a, b))
        e = g.kkt_error(0.0, q, v)
        assert np.isfinite(e).all() and abs(e[0] - o.kkt_error(0.0, q, v)) <= 1e-8 * max(1.0, e[0])


@pytest.mark.parametrize("N", [1, 2, 3])
def test_shortest_horizons_floating_base(N):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    T = 0.02 * N
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    q[7:] += 0.02 * np.cos(np.arange(12))
    fz = [0, 0, 0.25 * (-m.total_mass * m.gravity[2])]
    for Hip, Orc in ((HipOCP, OracleOCP), (HipParNMPC, OracleParNMPC)):
        par = Hip is HipParNMPC
        g, o = Hip(m, cost, cons, T, N, batch=2), Orc(m, cost, cons, T, N)
        for s in (g, o):
            s.set_contact_status([1, 1, 1, 1], pts)
            s.set_solution("q", ANYMAL_Q_STANDING)
            s.set_solution("v", np.zeros(m.nv))
            s.set_solution("f", fz)
            s.init(0.0) if par else s.init_constraints(0.0)
        assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
        for name in OCP_DIR:
            a, b = g.get(name, 1), o.get(name)
            assert a.shape == b.shape and rel_err(a, b) <= TOL, (N, par, name, rel_err(a, b))


def test_non_finite_inputs_do_not_hang_or_poison_other_instances():
    """A NaN in ONE instance's measured state: that instance's direction is not a number (the reference would print NaN KKT errors), the call comes
    back, the other instances of the batch are untouched -- bit for bit what they are without the poisoned neighbour."""
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    T, N, B = 1.0, 20, 5
    g, clean = HipUnOCP(m, cost, cons, T, N, batch=B), HipUnOCP(m, cost, cons, T, N, batch=B)
    q = np.tile(np.full(m.nv, 0.5), (B, 1)) + 0.01 * np.arange(B)[:, None]
    v = np.zeros((B, m.nv))
    for s in (g, clean):
        s.set_solution("q", np.full(m.nv, 0.5))
        s.set_solution("v", np.zeros(m.nv))
    qn = q.copy()
    qn[2, 3] = np.nan
    rc = g.update(0.0, qn, v)
    assert clean.update(0.0, q, v) == 0
    assert isinstance(rc, int)      # (an error code naming the instance, or 0 with NaNs there: it came back)
    assert not np.isfinite(g.direction("dq", 2)).all()
    for b in (0, 1, 3, 4):
        for name in UN_DIR:
            assert np.array_equal(g.direction(name, b), clean.direction(name, b)), (b, name)
    # floating base: inf in a weight -- the update returns (with an error code or with non-finite numbers), a fresh solver is unaffected
    ma = anymal_model()
    cost_a, cons_a = anymal_problem(ma, trotting_ref=False)
    bad = capi.Cost.from_buffer_copy(cost_a)
    bad.q_weight[7] = float("inf")
    pts = anymal_contact_points(ma)
    fz = [0, 0, 0.25 * (-ma.total_mass * ma.gravity[2])]
    for c in (bad, cost_a):
        h = HipOCP(ma, c, cons_a, 0.2, 8)
        h.set_contact_status([1, 1, 1, 1], pts)
        h.set_solution("q", ANYMAL_Q_STANDING)
        h.set_solution("v", np.zeros(ma.nv))
        h.set_solution("f", fz)
        h.init_constraints(0.0)
        rc = h.update(0.0, ANYMAL_Q_STANDING, np.zeros(ma.nv))
        if c is cost_a:
            assert rc == 0 and np.isfinite(h.get("dq")).all()
        else:
            assert rc != 0 or not np.isfinite(h.get("dq")).all()


def test_sizes_no_device_holds_are_refused_with_a_message():
    lib = capi.lib()
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    h = C.c_void_p()
    for N, batch in ((10 ** 7, 4096), (2 ** 30, 2 ** 20)):
        rc = lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, N, batch, 0, C.byref(h))
        assert rc != 0 and len(lib.idocp_last_error()) > 0, (N, batch)
    ma = anymal_model()
    cost_a, cons_a = anymal_problem(ma, trotting_ref=False)
    rc = lib.idocp_ocp_create(C.byref(ma), C.byref(cost_a), C.byref(cons_a), 1.0, 10 ** 6, 10 ** 5, 0, C.byref(h))
    assert rc != 0 and len(lib.idocp_last_error()) > 0
    # ... and the device is still there for the next one
    g = HipUnOCP(m, cost, cons, 1.0, 20)
    assert g.update(0.0, np.full(m.nv, 0.3), np.zeros(m.nv)) == 0


def test_create_destroy_cycles_do_not_leak_device_memory():
    """A service that builds a solver per request: two hundred create / use / destroy cycles of each handle type (and of clones) leave the device's
    free memory where it was."""
    import torch
    lib = capi.lib()
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    ma = anymal_model()
    cost_a, cons_a = anymal_problem(ma, trotting_ref=False)
    pts = anymal_contact_points(ma)

    def cycle():
        g = HipUnOCP(m, cost, cons, 1.0, 20, batch=8)
        assert g.update(0.0, np.full(m.nv, 0.3), np.zeros(m.nv)) == 0
        c = C.c_void_p()
        capi.check(lib.idocp_unocp_clone(g.h, C.byref(c)), "clone")
        lib.idocp_unocp_destroy(c)
        p = HipUnParNMPC(m, cost, cons, 1.0, 20, batch=4)
        p.init(0.0)
        h = HipOCP(ma, cost_a, cons_a, 0.2, 8, batch=4, max_num_impulse=2)
        h.set_contact_status([1, 1, 1, 1], pts)
        h.init_constraints(0.0)
        assert h.update(0.0, ANYMAL_Q_STANDING, np.zeros(ma.nv)) == 0
        n = HipParNMPC(ma, cost_a, cons_a, 0.2, 8, batch=2)
        n.set_contact_status([1, 1, 1, 1], pts)
        n.init(0.0)
        del g, p, h, n

    for _ in range(5):
        cycle()                                   # (allocator pools, code objects, the side streams' first use)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(200):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MB over 200 cycles" % ((free0 - free1) / 2 ** 20)


def test_handles_driven_from_concurrent_host_threads_give_the_single_threaded_results():
    """A service with a worker thread per request: four host threads, each with handles of its own (ctypes releases the interpreter lock inside
    every C-ABI call, so the calls really overlap), iterate different problems side by side.  Each must end exactly where it ends when it runs alone:
    the library keeps no state between handles (idocp_last_error is thread-local, every handle has its stream)."""
    import threading
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    ma = anymal_model()
    cost_a, cons_a = anymal_problem(ma, trotting_ref=False)
    pts = anymal_contact_points(ma)
    fz = [0, 0, 0.25 * (-ma.total_mass * ma.gravity[2])]

    def work(k, out):
        q = np.full(m.nv, 0.2 + 0.1 * k)
        g = HipUnOCP(m, cost, cons, 1.0, 20 + k, batch=1 + k)
        g.set_solution("q", q)
        qa = ANYMAL_Q_STANDING.copy()
        qa[7:] += 0.01 * (k + 1) * np.cos(np.arange(12))
        h = HipOCP(ma, cost_a, cons_a, 0.3, 10 + k, batch=1 + k)
        h.set_contact_status([1, 1, 1, 1], pts)
        h.set_solution("q", ANYMAL_Q_STANDING)
        h.set_solution("f", fz)
        h.init_constraints(0.0)
        for _ in range(15):
            assert g.update(0.0, q, np.zeros(m.nv)) == 0
            assert h.update(0.0, qa, np.zeros(ma.nv)) == 0
        out[k] = (g.solution("q", k).copy(), g.solution("lmd", k).copy(), h.get("q", k).copy(), h.get("u", k).copy(), g.kkt_error(0.0, q, np.zeros(m.nv))[k])

    alone, together = {}, {}
    for k in range(4):
        work(k, alone)
    threads = [threading.Thread(target=work, args=(k, together)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(together) == [0, 1, 2, 3]
    for k in range(4):
        for a, b in zip(alone[k], together[k]):
            assert np.array_equal(a, b), k
