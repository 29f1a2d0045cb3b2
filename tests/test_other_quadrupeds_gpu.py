"""Quadrupeds that are NOT ANYmal.  The reference's Robot is generic (robot.cpp:8-85); the contact-path kernels are instantiated for a free-flyer with four
serial three-joint legs and a point contact on each tip (ocp_capi.hip isQuadruped) and carry two forms of the leg sweeps: the general one (placement
rotation and axis of every joint from the model) and one specialised to ANYmal's pattern (identity placements, HAA about x, HFE / KFE about y).  The
other tests reach the general form only on ANYmal itself (IDOCP_GENERAL_AXES); here the MODEL differs: every leg joint gets a random placement rotation,
a random unit axis, a scaled offset; every body a random mass, centre of mass and inertia; every foot a rotated and shifted contact frame; the joint limits
move.  The oracle (generic in the model) is the checker: first Newton direction to 1e-10 stage by stage (the long double referee behind, helpers.parity),
OCPSolver with uniform contacts and along an event chain, ParNMPCSolver, and a few SQP iterations that must converge."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipOCP, HipParNMPC, OracleOCP, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem,
                     parity, rel_err)

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rotation(rng, angle):
    """A rotation by `angle` about a random axis (Rodrigues), row-major 9."""
    u = rng.normal(size=3)
    u /= np.linalg.norm(u)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def other_quadruped(seed):
    rng = np.random.default_rng(7000 + seed)
    m = anymal_model()
    for ji in range(1, m.njoints):
        R = np.array(m.plc_R[ji]).reshape(3, 3) @ rotation(rng, rng.uniform(0.1, 0.6))
        for k in range(9):
            m.plc_R[ji][k] = float(R.reshape(-1)[k])
        ax = np.array(m.axis[ji]) + rng.uniform(-0.6, 0.6, 3)
        ax /= np.linalg.norm(ax)
        for k in range(3):
            m.axis[ji][k] = float(ax[k])
            m.plc_p[ji][k] = float(m.plc_p[ji][k] * rng.uniform(0.7, 1.3) + rng.uniform(-0.02, 0.02))
    total = 0.0
    for ji in range(m.njoints):
        m.mass[ji] = float(m.mass[ji] * rng.uniform(0.5, 1.6))
        total += m.mass[ji]
        for k in range(3):
            m.com[ji][k] = float(m.com[ji][k] * rng.uniform(0.6, 1.4) + rng.uniform(-0.01, 0.01))
        # a positive definite inertia about the centre of mass with the principal axes turned
        I = np.array(m.inertia[ji]).reshape(3, 3)
        Q = rotation(rng, rng.uniform(0.0, 1.0))
        I = Q @ (I * rng.uniform(0.6, 1.5)) @ Q.T
        I = 0.5 * (I + I.T)
        assert np.linalg.eigvalsh(I).min() > 0
        for k in range(9):
            m.inertia[ji][k] = float(I.reshape(-1)[k])
    m.total_mass = total
    for c in range(m.ncontacts):
        R = np.array(m.contact_R[c]).reshape(3, 3) @ rotation(rng, rng.uniform(0.1, 0.8))
        for k in range(9):
            m.contact_R[c][k] = float(R.reshape(-1)[k])
        for k in range(3):
            m.contact_p[c][k] = float(m.contact_p[c][k] * rng.uniform(0.8, 1.2) + rng.uniform(-0.01, 0.01))
    for k in range(m.nu):                                     # (the limit arrays run over the actuated joints)
        m.q_min[k], m.q_max[k] = float(m.q_min[k] * rng.uniform(0.8, 1.2)), float(m.q_max[k] * rng.uniform(0.8, 1.2))
        m.v_max[k], m.u_max[k] = float(m.v_max[k] * rng.uniform(0.7, 1.3)), float(m.u_max[k] * rng.uniform(0.7, 1.3))
    return m, rng


def start(rng, m):
    q = ANYMAL_Q_STANDING.copy()
    q[7:] += rng.uniform(-0.1, 0.1, 12)
    quat = q[3:7] + rng.uniform(-0.05, 0.05, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    return q, rng.uniform(-0.2, 0.2, m.nv)


def prepare(solvers, m, par=False):
    for s in solvers:
        s.set_solution("q", ANYMAL_Q_STANDING)
        s.set_solution("v", np.zeros(m.nv))
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init(0.0) if par else s.init_constraints(0.0)


def test_the_models_leave_the_specialised_pattern():
    m, _ = other_quadruped(0)
    a = anymal_model()
    assert abs(np.linalg.norm(m.axis[1]) - 1) < 1e-14 and not np.allclose(m.axis[1], a.axis[1]) and not np.allclose(m.plc_R[5], a.plc_R[5])
    assert not np.allclose(m.contact_R[0], a.contact_R[0])


@pytest.mark.parametrize("seed", range(6))
def test_first_direction_with_uniform_contacts(seed):
    m, rng = other_quadruped(seed)
    cost, cons = anymal_problem(m, trotting_ref=False)
    N = int(rng.integers(4, 22))
    T = N * float(rng.uniform(0.01, 0.04))
    active = [[1, 1, 1, 1], [1, 0, 0, 1], [0, 0, 0, 0], [0, 1, 1, 1], [1, 0, 0, 0], [0, 1, 1, 0]][seed]
    par = seed % 3 == 1
    Hip, Orc = (HipParNMPC, OracleParNMPC) if par else (HipOCP, OracleOCP)
    g, o, h = Hip(m, cost, cons, T, N, batch=2), Orc(m, cost, cons, T, N), Orc(m, cost, cons, T, N, hp=True)
    pts = anymal_contact_points(m)
    for s in (g, o, h):
        s.set_contact_status(active, pts)
    prepare((g, o, h), m, par)
    q, v = start(rng, m)
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    ran = []

    def referee(name):
        if not ran:
            assert h.update(0.0, q, v) == 0
            ran.append(1)
        return h.get(name)

    worst = 0.0
    for name in OCP_DIR_FIELDS:
        worst = max(worst, parity(g.get(name, 1), o.get(name), lambda name=name: referee(name), (seed, active, N, name), tol=TOL, cap=1e-7))
    print("quadruped %d  contacts %s  N %d  %s  worst %.2e%s" % (seed, active, N, "ParNMPC" if par else "OCP", worst, "  (referee consulted)" if ran else ""))


@pytest.mark.parametrize("seed", range(4))
def test_first_direction_along_an_event_chain(seed):
    """A touch-down, a lift-off and a touch-down of other feet (impulse / aux / lift stages, switching constraint with 6 or 3 rows)."""
    m, rng = other_quadruped(10 + seed)
    cost, cons = anymal_problem(m, trotting_ref=False)
    N = int(rng.integers(14, 24))
    dt = float(rng.uniform(0.02, 0.04))
    T = N * dt
    E = 3
    g, o, h = (HipOCP(m, cost, cons, T, N, batch=2, max_num_impulse=E), OracleOCP(m, cost, cons, T, N, max_num_impulse=E),
               OracleOCP(m, cost, cons, T, N, max_num_impulse=E, hp=True))
    pts = anymal_contact_points(m)
    seq = [[1, 0, 0, 1], [1, 1, 1, 1], [0, 1, 1, 0], [0, 1, 1, 1]] if seed % 2 == 0 else [[1, 1, 1, 1], [0, 1, 1, 1], [1, 1, 1, 1], [1, 0, 0, 1]]
    times = (np.array([3, 8, 12]) + rng.uniform(0.2, 0.8, 3)) * dt
    for s in (g, o, h):
        s.set_contact_status(seq[0], pts)
        for nxt, t_ev in zip(seq[1:], times):
            s.push_back_contact_status(nxt, pts, float(t_ev))
    prepare((g, o, h), m)
    q, v = start(rng, m)
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    kinds = [c["kind"] for c in o.chain(0.0)]
    M = len(kinds)
    assert len(g.chain(0.0)) == M and (kinds.count("impulse"), kinds.count("lift")) == ((2, 1) if seed % 2 == 0 else (1, 2))
    ran = []

    def referee(name):
        if not ran:
            assert h.update(0.0, q, v) == 0
            ran.append(1)
        return h.get_chain(name, M)

    worst = 0.0
    for name in list(OCP_DIR_FIELDS) + ["dxi"]:
        worst = max(worst, parity(g.get_chain(name, M), o.get_chain(name, M), lambda name=name: referee(name), (seed, name), tol=TOL, cap=1e-6))
    print("quadruped %d  N %d  chain %d  worst %.2e%s" % (10 + seed, N, M, worst, "  (referee consulted)" if ran else ""))


def test_sqp_iterations_converge_and_follow_the_oracle():
    m, rng = other_quadruped(20)
    cost, cons = anymal_problem(m, trotting_ref=False)
    T, N = 0.5, 20
    g, o = HipOCP(m, cost, cons, T, N, batch=2), OracleOCP(m, cost, cons, T, N)
    pts = anymal_contact_points(m)
    for s in (g, o):
        s.set_contact_status([1, 1, 1, 1], pts)
    prepare((g, o), m)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    e0 = o.kkt_error(0.0, q, v)
    assert abs(g.kkt_error(0.0, q, v)[1] - e0) <= 1e-10 * e0
    for it in range(12):
        assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
        if it < 3:
            for f in ("q", "v", "a", "u", "f"):
                assert rel_err(g.get(f, 1), o.get(f)) < 1e-8, (it, f)
    e_g, e_o = g.kkt_error(0.0, q, v)[1], o.kkt_error(0.0, q, v)
    print("KKT error %.3e -> HIP %.3e, oracle %.3e" % (e0, e_g, e_o))
    assert e_g < 1e-3 * e0 and e_o < 1e-3 * e0
