"""GPU parity of TaskSpace3DCost / TaskSpace6DCost on a FLOATING-BASE robot (src/cost/task_space_3d_cost.cpp,
task_space_6d_cost.cpp: stage, terminal and impulse weights; the frame on a foot, on a thigh link, on the base) against the oracle:
OCPSolver on a uniform horizon and on a trotting chain with impulse stages, its line search, ParNMPCSolver on an event-free horizon.
The terms are evaluated outside the condensation kernel (idocp_amd/csrc/ocp_ext_kernel.hip)."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from idocp_amd.workloads import ANYMAL_URDF
from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, OCP_SOL_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, P,
                     anymal_contact_points, anymal_model, anymal_problem, parity, rel_err, trotting_sequence)

pytestmark = pytest.mark.gpu


def add_task(cost, frame_name, dim, ref_offset=(0.05, -0.03, 0.08)):
    lib = capi.lib()
    fid = lib.idocp_model_frame_id(ANYMAL_URDF.encode(), frame_name.encode())
    assert fid >= 0, frame_name
    joint = C.c_int()
    R, p = (C.c_double * 9)(), (C.c_double * 3)()
    capi.check(lib.idocp_model_frame_placement(ANYMAL_URDF.encode(), fid, C.byref(joint), R, p), "frame_placement")
    cost.task_dim = dim
    cost.task_joint = joint.value
    for k in range(9):
        cost.task_frame_R[k] = R[k]
    for k in range(3):
        cost.task_frame_p[k] = p[k]
    w = [30.0, 20.0, 40.0, 5.0, 6.0, 7.0] if dim == 6 else [30.0, 20.0, 40.0, 0, 0, 0]
    for k in range(6):
        cost.task_weight[k] = w[k]
        cost.task_weightf[k] = 2.0 * w[k]
        cost.task_weighti[k] = 0.5 * w[k]
    return joint.value, np.array(list(ref_offset))


def set_reference(cost, m, frame_pos, offset, yaw=0.2):
    c, s = np.cos(yaw), np.sin(yaw)
    Rref = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    for k in range(9):
        cost.task_ref[k] = Rref.flat[k]
    for k in range(3):
        cost.task_ref[9 + k] = frame_pos[k] + offset[k]


def start(solvers, m, seq=None):
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in solvers:
        if seq is None:
            s.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
        else:
            trotting_sequence(s, m, seq)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    return q, v


@pytest.mark.parametrize("frame,dim", [("LF_FOOT", 3), ("RH_THIGH", 3), ("base", 6), ("RF_SHANK", 6)])
def test_ocp_uniform_horizon(frame, dim):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    jf, off = add_task(cost, frame, dim)
    assert (jf == 0) == (frame == "base")
    pts = anymal_contact_points(m)
    pos = pts[0] if frame == "LF_FOOT" else np.array([0.0, 0.0, 0.45])      # (the reference only has to be near the frame)
    set_reference(cost, m, pos, off)
    o, g = OracleOCP(m, cost, cons, 0.5, 20), HipOCP(m, cost, cons, 0.5, 20)
    h = OracleOCP(m, cost, cons, 0.5, 20, hp=True)                      # long double referee
    q, v = start((o, g, h), m)
    o.init_constraints(0.0); g.init_constraints(0.0); h.init_constraints(0.0)
    rng = np.random.default_rng(11)
    q[7:] += 0.02 * rng.uniform(-1, 1, 12)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(25):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it <= 2:
            assert h.update(0.0, q, v) == 0
        if it in (0, 2):
            # 1e-10, or the referee's word (the Gauss-Newton task Hessian J^T W J with weights of 1e3 stiffens the stage blocks)
            for f in list(OCP_DIR_FIELDS) + list(OCP_SOL_FIELDS):
                parity(g.get(f), o.get(f), lambda f=f: h.get(f), (it, f), cap=1e-8)
    e_o2, e_g2 = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert e_g2 < 1e-6 * e_g and abs(np.log10(e_g2 / e_o2)) < 1.0


def chain_pair(dim, frame, batch=1, referee=False):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    add_task(cost, frame, dim)
    set_reference(cost, m, np.array([0.1, 0.0, 0.45]), (0.05, 0.0, 0.02))
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, batch=batch, max_num_impulse=nimp + 1)
    solvers = [o, g]
    h = None
    if referee:
        h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)      # long double referee
        solvers.append(h)
    q, v = start(solvers, m, seq=nimp)
    for s_ in solvers:
        s_.init_constraints(0.0)
    return (m, o, g, q, v, h) if referee else (m, o, g, q, v)


@pytest.mark.parametrize("frame,dim", [("base", 6), ("LH_FOOT", 3)])
def test_ocp_trotting_chain_with_impulse_and_terminal_weights(frame, dim):
    m, o, g, q, v, h = chain_pair(dim, frame, referee=True)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
        for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
            parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), (it, f), cap=1e-8)
    for f in ("q", "v", "a", "f", "u", "lmd", "gmm", "beta", "mu"):
        parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), f, cap=1e-8)


def test_ocp_line_search_cost():
    m, o, g, q, v = chain_pair(6, "base", batch=2)
    o.lib.oracle_ocp_cost_and_violation.argtypes = [C.c_void_p, C.c_double, capi.c_double_p]
    o.lib.oracle_ocp_compute_direction.argtypes = [C.c_void_p, C.c_double, capi.c_double_p, capi.c_double_p]
    q[7:] += 0.05
    assert o.lib.oracle_ocp_compute_direction(o.h, 0.0, P(q), P(v)) == 0
    capi.check(g.lib.idocp_ocp_compute_direction(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))), "compute_direction")
    ap, _ = g.step_sizes()
    for alpha in (0.0, 0.01 * ap[0], 0.5 * ap[0], ap[0]):
        ref = np.zeros(2)
        assert o.lib.oracle_ocp_cost_and_violation(o.h, alpha, P(ref)) == 0
        c, vi = np.zeros(g.batch), np.zeros(g.batch)
        capi.check(g.lib.idocp_ocp_line_search_eval(g.h, P(np.full(g.batch, alpha)), P(c), P(vi)), "line_search_eval")
        assert abs(c[0] - ref[0]) <= 1e-10 * max(1.0, abs(ref[0])), (alpha, c[0], ref[0])
        assert abs(vi[0] - ref[1]) <= 1e-10 * max(1.0, abs(ref[1])), (alpha, vi[0], ref[1])


def test_parnmpc_event_free_horizon():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    add_task(cost, "RH_THIGH", 6)
    set_reference(cost, m, np.array([-0.3, -0.1, 0.45]), (0.02, 0.0, 0.03))
    o, g = OracleParNMPC(m, cost, cons, 0.5, 20), HipParNMPC(m, cost, cons, 0.5, 20)
    hp = OracleParNMPC(m, cost, cons, 0.5, 20, hp=True)                   # long double referee
    q, v = start((o, g, hp), m)
    o.init(0.0); g.init(0.0); hp.init(0.0)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and hp.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        parity(g.get(f), o.get(f), lambda f=f: hp.get(f), f, cap=5e-8)


def moving_reference(times, dim, p0=(0.1, 0.0, 0.47), vel=(0.25, -0.1, 0.05), t0=0.2, tf=1.2, yaw_rate=0.4):
    """A reference object of the kind the reference's examples pass to TimeVaryingTaskSpace3DCost / 6DCost: a pose that moves with
    constant velocity (and yaw rate) inside [t0, tf] and rests outside; tabulated at the given times as [M][12]."""
    refs = np.zeros((len(times), 12))
    for k, t in enumerate(times):
        tau = min(max(t, t0), tf) - t0
        yaw = yaw_rate * tau if dim == 6 else 0.0
        c, s_ = np.cos(yaw), np.sin(yaw)
        refs[k, :9] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]]).ravel()
        refs[k, 9:] = np.array(p0) + np.array(vel) * tau
    return refs


@pytest.mark.parametrize("frame,dim", [("base", 6), ("LH_FOOT", 3)])
def test_time_varying_reference_on_a_trotting_chain(frame, dim):
    """TimeVaryingTaskSpace3DCost / TimeVaryingTaskSpace6DCost on the floating base (round 4; src/cost/time_varying_task_space_3d_cost.cpp,
    time_varying_task_space_6d_cost.cpp): the reference pose is evaluated at the time of EVERY stage of the chain -- grid stages, the
    impulse / aux / lift stages at their event times, the terminal stage -- through idocp_ocp_get_chain_times / idocp_ocp_set_task_refs."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    add_task(cost, frame, dim)
    cost.task_time_varying = 1
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)
    g = HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=nimp + 1)
    q, v = start((o, g, h), m, seq=nimp)
    # without poses the solver refuses to run (the device cannot call the reference object)
    assert g.lib.idocp_ocp_init_constraints(g.h, 0.0) != 0
    times = g.chain_times(0.0)
    chain_o = o.chain(0.0)
    assert len(times) == len(chain_o) and np.allclose(times, [c["t"] for c in chain_o], rtol=0, atol=1e-12)
    assert np.sum(np.diff(times) == 0.0) >= nimp           # an impulse stage and its aux stage share the event time
    refs = moving_reference(times, dim)
    assert np.abs(refs[0, 9:] - refs[-1, 9:]).max() > 0.1   # the reference does move along the horizon
    g.set_task_refs(0.0, refs)
    o.set_task_refs(times, refs); h.set_task_refs(times, refs)
    for s_ in (o, g, h):
        s_.init_constraints(0.0)
    M = len(times)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) < 1e-10 * max(1.0, e_o) and e_g[0] == e_g[1]
    for it in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
        for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
            parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), (it, f), cap=1e-8)
    # the constant-reference solver of the same problem moves differently: the table is what the kernels read
    cost.task_time_varying = 0
    for k in range(12):
        cost.task_ref[k] = refs[0, k]
    gc = HipOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    start((gc,), m, seq=nimp)
    gc.init_constraints(0.0)
    assert gc.update(0.0, q, v) == 0
    assert np.abs(gc.get_chain("dq", M) - g.get_chain("dq", M)).max() > 1e-6
    # a call at another time needs poses for that chain
    assert g.update(0.05, q, v) != 0
    t2 = g.chain_times(0.05)
    g.set_task_refs(0.05, moving_reference(t2, dim))
    assert g.update(0.05, q, v) == 0


def test_time_varying_reference_parnmpc_event_free():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    add_task(cost, "RH_THIGH", 3)
    cost.task_time_varying = 1
    o, g = OracleParNMPC(m, cost, cons, 0.5, 20), HipParNMPC(m, cost, cons, 0.5, 20)
    hp = OracleParNMPC(m, cost, cons, 0.5, 20, hp=True)
    q, v = start((o, g, hp), m)
    times = g.chain_times(0.0)
    assert np.allclose(times[:20], [c["t"] for c in o.chain(0.0)][:20], rtol=0, atol=1e-12)
    refs = moving_reference(times, 3, p0=(-0.3, -0.1, 0.45), t0=0.1, tf=0.4)
    g.set_task_refs(0.0, refs)
    o.set_task_refs(times, refs); hp.set_task_refs(times, refs)
    o.init(0.0); g.init(0.0); hp.init(0.0)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and hp.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        parity(g.get(f), o.get(f), lambda f=f: hp.get(f), f, cap=5e-8)


@pytest.mark.parametrize("frame,dim,time_varying", [("base", 6, False), ("LF_FOOT", 3, True)])
def test_parnmpc_chain_with_events(frame, dim, time_varying):
    """The task-space cost on a ParNMPC horizon WITH discrete events: the impulse stage takes the cost with its impulse weights
    (ImpulseSplitParNMPC::linearizeOCP -> CostFunction::computeImpulseCostDerivatives / Hessian, impulse_split_parnmpc.hxx:60-112;
    task_space_{3d,6d}_cost.cpp impulse members), the aux / lift stages the stage cost with their own dt; the reference may move with
    the stage's time.  KKT error, first direction along the chain (the long double referee decides where an event pair sits), the
    line search's cost and the accepted iterates."""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    add_task(cost, frame, dim)
    pos = np.array([0.0, 0.0, 0.48]) if frame == "base" else np.array([0.35, 0.2, 0.02])
    set_reference(cost, m, pos, (0.03, -0.02, 0.04))
    cost.task_time_varying = 1 if time_varying else 0
    events = [([0, 1, 1, 0], 0.52), ([1, 1, 1, 1], 0.83)]
    mk = lambda cls, **kw: cls(m, cost, cons, 1.0, 20, max_num_impulse=3, **kw)
    o, g, hp = mk(OracleParNMPC), mk(HipParNMPC, batch=2), mk(OracleParNMPC, hp=True)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s_ in (o, g, hp):
        pts = anymal_contact_points(m).copy()
        s_.set_contact_status([1, 1, 1, 1], pts)
        for status, t_ev in events:
            s_.push_back_contact_status(status, pts, t_ev)
        s_.set_solution("q", q)
        s_.set_solution("v", v)
        s_.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    co = o.chain(0.0)
    M = len(co)
    kinds = [c["kind"] for c in co]
    assert "impulse" in kinds
    if time_varying:
        times = g.chain_times(0.0)
        assert np.allclose(times[:M], [c["t"] for c in co], rtol=0, atol=1e-12)
        refs = moving_reference(times, dim, p0=tuple(pos), t0=0.1, tf=0.9)
        g.set_task_refs(0.0, refs)
        o.set_task_refs(times, refs); hp.set_task_refs(times, refs)
    o.init(0.0); g.init(0.0); hp.init(0.0)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert o.lib.oracle_parnmpc_compute_direction(o.h, 0.0, P(q), P(v)) == 0
    assert g.lib.idocp_parnmpc_compute_direction(g.h, 0.0, P(g._bc(q, g.nq)), P(g._bc(v, g.nv))) == 0
    ap, _ = g.step_sizes()
    for alpha in (0.0, 0.4 * ap[0], ap[0]):
        ref = np.zeros(2)
        assert o.lib.oracle_parnmpc_cost_and_violation(o.h, alpha, P(q), P(v), P(ref)) == 0
        c, vi = np.zeros(g.batch), np.zeros(g.batch)
        assert g.lib.idocp_ocp_line_search_eval(g.h, P(np.full(g.batch, alpha)), P(c), P(vi)) == 0
        assert abs(c[0] - ref[0]) <= 1e-8 * max(1.0, abs(ref[0])), (alpha, c[0], ref[0])
        assert abs(vi[0] - ref[1]) <= 1e-8 * max(1.0, abs(ref[1])), (alpha, vi[0], ref[1])
        assert c[0] == c[-1]
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and hp.update(0.0, q, v) == 0
    keep_reg = np.array([k != "impulse" for k in kinds])
    for f in OCP_DIR_FIELDS:
        keep = keep_reg if f in ("du", "dnu_passive") else np.ones(M, bool)
        parity(g.get_chain(f, M + 1)[:M][keep], o.get_chain(f, M)[keep], lambda f=f, keep=keep: hp.get_chain(f, M)[keep], f, cap=1e-7)
    for it in range(2):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in ("q", "v", "a", "f"):
        assert rel_err(g.get_chain(f, M + 1)[:M], o.get_chain(f, M)) < 1e-6, f
