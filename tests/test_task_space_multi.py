"""More than one task-space cost component per problem (round 6).  The reference's CostFunction takes any number of components
(include/idocp/cost/cost_function.hpp:67 push_back; cost_function.hxx sums them); the flat cost block carries the first one in its task_*
fields and up to three more in task_extra (idocp_hip.h), each a TaskSpace3DCost / TaskSpace6DCost on a frame of its own.

* CPU: the oracle's terms of a two- / three-component cost are the SUM of the one-component costs' terms (cost, gradient, Gauss-Newton Hessian),
  and the gradient is the derivative of the cost (central differences);
* GPU: UnOCPSolver (iiwa14: 6D on the end effector + 3D on an elbow link + 3D on the wrist) -- Newton direction, Riccati factorisation,
  KKT error and line-search cost against the oracle at 1e-10; OCPSolver (ANYmal: 3D on a foot + 6D on the base + 3D on a thigh) on a uniform
  horizon and on a trotting chain with impulse stages; ParNMPCSolver event-free;
* the facade: two TaskSpace3DCost components pushed into one CostFunction land in task_* and task_extra."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from idocp_amd import capi
from idocp_amd.workloads import ANYMAL_URDF, IIWA_URDF, task_space_problem
from helpers import (ANYMAL_Q_STANDING, DIR_FIELDS, OCP_DIR_FIELDS, OCP_SOL_FIELDS, SOL_FIELDS, HipOCP, HipParNMPC, HipUnOCP, OracleOCP, OracleParNMPC,
                     OracleUnOCP, P, ROOT, anymal_contact_points, anymal_model, anymal_problem, iiwa14_model, parity, rel_err, trotting_sequence)

Q0 = np.array([0, np.pi / 2, 0, np.pi / 2, 0, np.pi / 2, 0.0])


def frame_of(urdf, name_or_id):
    lib = capi.lib()
    fid = name_or_id if isinstance(name_or_id, int) else lib.idocp_model_frame_id(urdf.encode(), name_or_id.encode())
    assert fid >= 0, name_or_id
    joint = C.c_int()
    R, p = (C.c_double * 9)(), (C.c_double * 3)()
    capi.check(lib.idocp_model_frame_placement(urdf.encode(), fid, C.byref(joint), R, p), "frame_placement")
    return joint.value, list(R), list(p)


def iiwa_cost(ncomp):
    """6D on the end effector (the example's cost) + 3D on iiwa_link_4 + 3D on iiwa_link_6, constant references"""
    m = iiwa14_model()
    cost, cons = task_space_problem(m, dim=6, weight=300.0)
    extras = [("iiwa_link_4", 3, [40.0, 60.0, 50.0], [0.25, 0.05, 0.75]), ("iiwa_link_6", 3, [15.0, 25.0, 35.0], [0.45, -0.05, 0.85])]
    for name, dim, w, pref in extras[:ncomp - 1]:
        j, R, p = frame_of(IIWA_URDF, name)
        cost.add_task(dim, j, R, p, w + [0, 0, 0], [2 * x for x in w] + [0, 0, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1] + pref)
    return m, cost, cons


def only_component(cost_all, which):
    """a cost block that carries ONLY component `which` (0 = the task_* block) of cost_all"""
    c = capi.Cost()
    C.memmove(C.byref(c), C.byref(cost_all), C.sizeof(capi.Cost))
    if which > 0:
        t = cost_all.task_extra[which - 1]
        c.task_dim, c.task_joint = t.dim, t.joint
        for name, src in (("task_frame_R", t.frame_R), ("task_frame_p", t.frame_p), ("task_weight", t.weight), ("task_weightf", t.weightf),
                          ("task_weighti", t.weighti), ("task_ref", t.ref)):
            dst = getattr(c, name)
            for k in range(len(dst)):
                dst[k] = src[k]
    c.task_extra_count = 0
    return c


@pytest.mark.parametrize("ncomp", [2, 3])
def test_oracle_terms_of_several_components_are_the_sum_of_the_single_ones(ncomp):
    m, cost, cons = iiwa_cost(ncomp)
    N, T = 6, 0.3
    o = OracleUnOCP(m, cost, cons, T, N)
    singles = [OracleUnOCP(m, only_component(cost, k), cons, T, N) for k in range(ncomp)]
    rng = np.random.default_rng(5)
    for stage in (0, 3, N):
        q = Q0 + 0.3 * rng.uniform(-1, 1, 7)
        c, g, H = o.task_terms(stage, q)
        cs, gs, Hs = zip(*[s.task_terms(stage, q) for s in singles])
        assert abs(c - sum(cs)) < 1e-12 * max(1.0, abs(c))
        assert np.max(np.abs(g - sum(gs))) < 1e-12 * max(1.0, np.max(np.abs(g)))
        assert np.max(np.abs(H - sum(Hs))) < 1e-12 * max(1.0, np.max(np.abs(H)))
        assert all(abs(x) > 1e-6 for x in cs), "every component contributes"
        # gradient = derivative of the cost (central differences)
        eps = 1e-6
        fd = np.array([(o.task_terms(stage, q + eps * e)[0] - o.task_terms(stage, q - eps * e)[0]) / (2 * eps) for e in np.eye(7)])
        assert np.max(np.abs(fd - g)) < 1e-6 * max(1.0, np.max(np.abs(g)))


@pytest.mark.gpu
@pytest.mark.parametrize("ncomp", [2, 3])
def test_unocp_direction_riccati_kkt_error_and_line_search(ncomp):
    m, cost, cons = iiwa_cost(ncomp)
    N, T = 20, 1.0
    o, g = OracleUnOCP(m, cost, cons, T, N), HipUnOCP(m, cost, cons, T, N, batch=2)
    for s in (o, g):
        s.set_solution("q", Q0)
        s.set_solution("v", np.zeros(7))
    v0 = np.zeros(7)
    eo, eg = o.kkt_error(0.0, Q0, v0), g.kkt_error(0.0, Q0, v0)[0]
    assert abs(eg - eo) < 1e-10 * max(1.0, eo)
    # the further components change the problem (a regression that dropped them would still match a one-component oracle)
    o1 = OracleUnOCP(m, only_component(cost, 0), cons, T, N)
    o1.set_solution("q", Q0); o1.set_solution("v", np.zeros(7))
    assert abs(o1.kkt_error(0.0, Q0, v0) - eo) > 1e-3 * eo
    for it in range(3):
        assert o.update(0.0, Q0, v0) == 0 and g.update(0.0, Q0, v0) == 0
        for f in DIR_FIELDS:
            assert rel_err(g.direction(f, 1), o.direction(f)) < (1e-10 if it == 0 else 1e-8), (it, f)
        if it == 0:
            Po, so, Ko, ko = o.riccati()
            Pg, sg, Kg, kg = g.riccati()
            assert rel_err(Pg, Po) < 1e-10 and rel_err(sg, so) < 1e-10 and rel_err(Kg, Ko) < 1e-10 and rel_err(kg, ko) < 1e-10
            for f in SOL_FIELDS:
                assert rel_err(g.solution(f), o.solution(f)) < 1e-10, f
        eo, eg = o.kkt_error(0.0, Q0, v0), g.kkt_error(0.0, Q0, v0)[0]
        assert abs(eg - eo) < 1e-8 * max(1.0, eo), it
    # filter line search: cost and violation of trial steps
    for alpha in (0.0, 0.3, 1.0):
        ref = o.cost_and_violation(alpha)
        cg, vg = g.cost_and_violation(alpha)
        # (the iterates of the two sides agree to 1e-8 after three iterations, and log6 of a small pose error loses digits in the cost:
        #  tests/test_task_space_gpu.py::test_line_search_cost_includes_the_task_terms)
        assert abs(cg[0] - ref[0]) <= 1e-7 * max(1.0, abs(ref[0])), (alpha, cg[0], ref[0])
        assert abs(vg[0] - ref[1]) <= 1e-8 * max(1.0, abs(ref[1])), (alpha, vg[0], ref[1])


def anymal_cost(ncomp, trotting_ref):
    """3D on the LF foot + 6D on the base + 3D on the RH thigh"""
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=trotting_ref)
    pts = anymal_contact_points(m)
    j, R, p = frame_of(ANYMAL_URDF, "LF_FOOT")
    cost.task_dim, cost.task_joint = 3, j
    for k in range(9):
        cost.task_frame_R[k] = R[k]
    for k in range(3):
        cost.task_frame_p[k] = p[k]
    w = [30.0, 20.0, 40.0, 0, 0, 0]
    for k in range(6):
        cost.task_weight[k], cost.task_weightf[k], cost.task_weighti[k] = w[k], 2 * w[k], 0.5 * w[k]
    for k, x in enumerate([1, 0, 0, 0, 1, 0, 0, 0, 1] + list(pts[0] + np.array([0.05, -0.03, 0.08]))):
        cost.task_ref[k] = x
    c, s = np.cos(0.2), np.sin(0.2)
    extras = [("base", 6, [9.0, 8.0, 7.0, 25.0, 35.0, 45.0], [c, -s, 0, s, c, 0, 0, 0, 1, 0.03, -0.02, 0.5]),
              ("RH_THIGH", 3, [12.0, 14.0, 16.0, 0, 0, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1, -0.3, -0.2, 0.35])]
    for name, dim, wv, ref in extras[:ncomp - 1]:
        j, R, p = frame_of(ANYMAL_URDF, name)
        cost.add_task(dim, j, R, p, wv, [2 * x for x in wv], ref, weighti=[0.5 * x for x in wv])
    return m, cost, cons


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


@pytest.mark.gpu
@pytest.mark.parametrize("ncomp", [2, 3])
def test_ocp_uniform_horizon(ncomp):
    m, cost, cons = anymal_cost(ncomp, False)
    o, g = OracleOCP(m, cost, cons, 0.5, 20), HipOCP(m, cost, cons, 0.5, 20, batch=2)
    h = OracleOCP(m, cost, cons, 0.5, 20, hp=True)
    o1 = OracleOCP(m, only_component(cost, 0), cons, 0.5, 20)
    q, v = start((o, g, h, o1), m)
    for s in (o, g, h, o1):
        s.init_constraints(0.0)
    q[7:] += 0.02 * np.random.default_rng(11).uniform(-1, 1, 12)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    assert abs(o1.kkt_error(0.0, q, v) - e_o) > 1e-3 * e_o                # the further components are part of the problem
    for it in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
        for f in list(OCP_DIR_FIELDS) + list(OCP_SOL_FIELDS):
            parity(g.get(f, 1), o.get(f), lambda f=f: h.get(f), (it, f), cap=1e-8)
    # line search
    o.lib.oracle_ocp_cost_and_violation.argtypes = [C.c_void_p, C.c_double, capi.c_double_p]
    ap, _ = g.step_sizes()
    for alpha in (0.0, 0.5 * ap[0]):
        ref = np.zeros(2)
        assert o.lib.oracle_ocp_cost_and_violation(o.h, alpha, P(ref)) == 0
        c, vi = np.zeros(g.batch), np.zeros(g.batch)
        capi.check(g.lib.idocp_ocp_line_search_eval(g.h, P(np.full(g.batch, alpha)), P(c), P(vi)), "line_search_eval")
        assert abs(c[0] - ref[0]) <= 1e-9 * max(1.0, abs(ref[0])), (alpha, c[0], ref[0])


@pytest.mark.gpu
def test_ocp_trotting_chain_with_impulse_stages():
    m, cost, cons = anymal_cost(3, True)
    N, T, nimp = 31, 1.55, 2
    o = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    g = HipOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
    h = OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1, hp=True)
    q, v = start((o, g, h), m, seq=nimp)
    for s in (o, g, h):
        s.init_constraints(0.0)
    M = len(o.chain(0.0))
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-10 * max(1.0, e_o)
    for it in range(2):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0 and h.update(0.0, q, v) == 0
        for f in ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu"):
            parity(g.get_chain(f, M), o.get_chain(f, M), lambda f=f: h.get_chain(f, M), (it, f), cap=1e-8)


@pytest.mark.gpu
def test_parnmpc_event_free_horizon():
    m, cost, cons = anymal_cost(2, False)
    N, T = 20, 0.5
    o, g = OracleParNMPC(m, cost, cons, T, N), HipParNMPC(m, cost, cons, T, N)
    q, v = start((o, g), m)
    o.init(0.0); g.init(0.0)
    q[7:] += 0.03
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)[0]
    assert abs(e_g - e_o) < 1e-9 * max(1.0, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in ("dq", "dv", "da", "du", "dlmd", "dgmm"):
        assert rel_err(g.get(f), o.get(f)) < 1e-9, f


@pytest.mark.gpu
def test_argument_errors():
    m, cost, cons = iiwa_cost(2)
    lib = capi.lib()
    h = C.c_void_p()
    cost.task_extra[0].dim = 4
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, 1, 0, C.byref(h)) == -1
    assert b"task_extra[0].dim" in lib.idocp_last_error()
    cost.task_extra[0].dim = 3
    cost.task_extra_count = 4
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, 1, 0, C.byref(h)) == -1
    cost.task_extra_count = 1
    cost.task_dim = 0
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 10, 1, 0, C.byref(h)) == -1


def test_facade_pushes_further_components_into_task_extra(tmp_path):
    src = tmp_path / "multi.cpp"
    src.write_text(r'''
#include <cstdio>
#include <memory>
#include "idocp/robot/robot.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/cost/task_space_cost.hpp"
int main(int argc, char** argv) {
  idocp::Robot robot(argv[1]);
  auto cost = std::make_shared<idocp::CostFunction>();
  auto cs = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  auto t6 = std::make_shared<idocp::TaskSpace6DCost>(robot, 22);
  t6->set_q_6d_weight(Eigen::Vector3d::Constant(300), Eigen::Vector3d::Constant(200));
  auto t3 = std::make_shared<idocp::TaskSpace3DCost>(robot, 10);
  Eigen::Vector3d w; w << 40, 60, 50;
  t3->set_q_3d_weight(w);
  Eigen::Vector3d r; r << 0.25, 0.05, 0.75;
  t3->set_q_3d_ref(r);
  cost->push_back(t6);
  cost->push_back(t3);
  cost->push_back(cs);                       // (pushed last: must not wipe the task components)
  const idocp_cost_t& c = cost->native();
  std::printf("%d %d %d %d %g %g %g %g\n", c.task_dim, c.task_extra_count, c.task_extra[0].dim, c.task_extra[0].joint, c.task_extra[0].weight[1],
              c.task_extra[0].ref[9], c.task_weight[0], c.task_weight[3]);
  return 0;
}
''')
    exe = tmp_path / "multi"
    lib = os.path.join(ROOT, "idocp_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + lib, "-lidocp_hip", "-Wl,-rpath," + lib],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe), IIWA_URDF], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    vals = out.stdout.split()
    assert vals[:3] == ["6", "1", "3"], out.stdout
    assert float(vals[4]) == 60.0 and float(vals[5]) == 0.25 and float(vals[6]) == 200.0 and float(vals[7]) == 300.0
