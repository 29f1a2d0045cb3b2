"""idocp::OCPSolver / idocp::ParNMPCSolver on a fixed-base robot without contact frames (the reference's examples/iiwa14/ocp_benchmark.cpp
and parnmpc_benchmark.cpp).  The facade binds them to the fixed-base kernels (include/idocp/ocp/ocp_solver.hpp, parnmpc_solver.hpp); here
those kernels are held, through the C ABI, to the oracle's restatement of the CONTACT-PATH solvers on that robot (oracle/ocp.cpp with no
passive rows: the reference's `has_floating_base_ == false` branches) -- Newton direction to 1e-10 stage by stage, step sizes, KKT error,
the torque feedback gain, and the setter that leaves the slack / dual variables alone.  The example drivers run the facade end to end."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import (HipUnOCP, HipUnParNMPC, IIWA_URDF, OracleOCP, OracleParNMPC, P, arr, iiwa14_model, rel_err, unocp_problem)
from idocp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TOL = 1e-10                                                  # north_star: direction to 1e-10 FP64
DIR = ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta")
SOL = ("q", "v", "a", "u", "lmd", "gmm", "beta")


def start(model):
    return np.full(model.nq, 2.0), np.zeros(model.nv)       # examples/iiwa14/ocp_benchmark.cpp:44-45


@pytest.mark.parametrize("N", [20, 100])
def test_fixed_base_ocp_solver_direction_matches_the_contact_path_restatement(N):
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    T = 1.0 if N == 20 else 5.0
    q, v = start(model)
    g, o = HipUnOCP(model, cost, cons, T, N), OracleOCP(model, cost, cons, T, N)
    for s in (g, o):
        s.set_solution("q", q)
        s.set_solution("v", v)
    o.init_constraints(0.0)
    e_g, e_o = g.kkt_error(0.0, q, v)[0], o.kkt_error(0.0, q, v)
    assert abs(e_g - e_o) <= 1e-10 * e_o
    # iteration 0 is the parity statement (both linearise the same iterate); the next two follow the drift of the ITERATION, which amplifies
    # rounding differences on this problem (tests/test_oracle_fixed_base.py: the two CPU restatements drift apart the same way)
    for it, tol in enumerate((TOL, 1e-8, 1e-8)):
        assert g.update(0.0, q, v) == 0
        assert o.update(0.0, q, v) == 0
        for name in DIR:
            a, b = g.direction(name), o.get(name)
            assert a.shape == b.shape, name
            assert rel_err(a, b) <= tol, (N, it, name, rel_err(a, b))
        (pg, dg), (po, do) = g.step_sizes(), o.step_sizes()
        assert abs(pg[0] - po) <= 1e3 * tol * po and abs(dg[0] - do) <= 1e3 * tol * do
        for name in SOL:
            assert rel_err(g.solution(name), o.get(name)) <= 10 * tol, (N, it, name)


def test_torque_feedback_gain_matches_the_riccati_gain_of_the_contact_path_restatement():
    """OCPSolver::getStateFeedbackGain (ocp_solver.cpp:101-111) through idocp_unocp_get_torque_feedback_gain: the oracle's OCPSolver computes
    K from ITS Riccati recursion on (x, u); the product maps the acceleration gain through dID/dq, dID/dv, M of the same linearisation."""
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    T, N, nv = 1.0, 20, model.nv
    q, v = start(model)
    g, o = HipUnOCP(model, cost, cons, T, N), OracleOCP(model, cost, cons, T, N)
    for s in (g, o):
        s.set_solution("q", q)
        s.set_solution("v", v)
    o.init_constraints(0.0)
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    _, _, K, _ = o.riccati()                                 # [N][nu][2 nv]
    for i in (0, 7, N - 1):
        Kq, Kv = np.zeros((nv, nv)), np.zeros((nv, nv))
        capi.check(g.lib.idocp_unocp_get_torque_feedback_gain(g.h, 0, i, P(Kq), P(Kv)), "gain")
        got = np.hstack([Kq.T, Kv.T])                        # col-major -> [row, col]
        assert np.abs(got - K[i]).max() <= 1e-9 * max(1.0, np.abs(K[i]).max()), i
    assert g.lib.idocp_unocp_get_torque_feedback_gain(g.h, 0, N, P(Kq), P(Kv)) != 0          # the terminal stage has no policy


def test_set_solution_only_leaves_slack_and_dual_alone():
    """OCPSolver::setSolution does not touch the constraints (ocp_solver.cpp:116-165), UnOCPSolver::setSolution re-initialises them
    (unocp_solver.cpp:180)."""
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    g = HipUnOCP(model, cost, cons, 1.0, 20)
    q, v = start(model)
    g.set_solution("q", q)
    sl0, du0 = g.constraint_data()
    capi.check(g.lib.idocp_unocp_set_solution_only(g.h, b"q", P(arr(np.full(model.nq, 1.0)))), "set_solution_only")
    assert np.allclose(g.solution("q"), 1.0)
    sl1, du1 = g.constraint_data()
    assert np.array_equal(sl0, sl1) and np.array_equal(du0, du1)
    g.set_solution("q", np.full(model.nq, 1.0))
    sl2, _ = g.constraint_data()
    assert not np.array_equal(sl0, sl2)
    assert g.lib.idocp_unocp_set_solution_only(g.h, b"f", P(arr(np.zeros(3)))) != 0


def test_fixed_base_parnmpc_solver_matches_the_contact_path_restatement():
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    T, N = 1.0, 20
    q, v = start(model)
    g, o = HipUnParNMPC(model, cost, cons, T, N), OracleParNMPC(model, cost, cons, T, N)
    for s in (g, o):
        s.set_solution("q", q)
        s.set_solution("v", v)
    g.init(0.0)
    o.init(0.0)
    e_g, e_o = g.kkt_error(0.0, q, v)[0], o.kkt_error(0.0, q, v)
    assert abs(e_g - e_o) <= 1e-10 * e_o
    assert g.update(0.0, q, v) == 0 and o.update(0.0, q, v) == 0
    for name in DIR + SOL:
        a, b = g.get(name), o.get(name)
        assert a.shape == b.shape, name
        assert rel_err(a, b) <= TOL, (name, rel_err(a, b))
    (pg, dg), (po, do) = g.step_sizes(), o.step_sizes()
    assert abs(pg[0] - po) <= 1e-9 * po and abs(dg[0] - do) <= 1e-9 * do


def kkt_errors(stdout):
    init = float(re.search(r"Initial KKT error = ([-+0-9.eE]+|nan|inf)", stdout).group(1))
    its = [float(x) for x in re.findall(r"KKT error after iteration \d+ = ([-+0-9.eE]+|nan|inf)", stdout)]
    return init, its


def run_example(name, *args):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), name], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([os.path.join(ROOT, "examples", name), IIWA_URDF, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return r.stdout


def test_iiwa14_ocp_benchmark_example_follows_the_contact_path_restatement():
    out = run_example("iiwa14_ocp_benchmark", "20")
    init, its = kkt_errors(out)
    assert len(its) == 50 and "CPU time per update" in out
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    q, v = start(model)
    o = OracleOCP(model, cost, cons, 1.0, 20)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init_constraints(0.0)
    ref = o.kkt_error(0.0, q, v)
    assert abs(init - ref) <= 1e-5 * ref                      # (six printed digits)
    for k in range(3):                                       # (later iterations: the iteration amplifies rounding differences, see above)
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 1e-4 * ref, (k, its[k], ref)
    assert its[-1] < 1e-3 * init
    # the gain printed at the end is the torque policy of stage 0 of the LAST linearisation; finite and not the acceleration gain's scale
    m = re.search(r"du/dq\(0, 0\) of the policy at stage 0: ([-+0-9.eE]+), du/dv\(0, 0\): ([-+0-9.eE]+)", out)
    assert m and np.isfinite(float(m.group(1))) and np.isfinite(float(m.group(2)))


def test_iiwa14_parnmpc_benchmark_example_follows_the_contact_path_restatement():
    out = run_example("iiwa14_parnmpc_benchmark", "20")
    init, its = kkt_errors(out)
    assert len(its) == 100 and "feasible: yes" in out
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    q, v = start(model)
    o = OracleParNMPC(model, cost, cons, 1.0, 20)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init(0.0)
    ref = o.kkt_error(0.0, q, v)
    assert abs(init - ref) <= 1e-5 * ref
    for k in range(3):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 1e-4 * ref, (k, its[k], ref)
    assert its[-1] < 1e-3 * init


def test_iiwa14_config_space_ocp_example_converges():
    """BASELINE.json configs[0] (examples/iiwa14/config_space_ocp.cpp: T = 3, N = 60, 50 Nm, pi / 2 rad/s) through the facade."""
    out = run_example("iiwa14_config_space_ocp")
    init, its = kkt_errors(out)
    assert len(its) == 30 and its[-1] < 1e-6 * init and "feasible: yes" in out
