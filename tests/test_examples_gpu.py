"""The C++ facade (include/idocp/**) end to end on the GPU: build the example drivers with g++,
run them, and compare the KKT errors they print (ocpbenchmarker::Convergence,
include/idocp/utils/ocp_benchmarker.hxx:36-52) with the oracle driven on the same problem."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import ANYMAL_Q_STANDING, ANYMAL_URDF, IIWA_URDF, OracleOCP, anymal_contact_points, anymal_model
from idocp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build_examples():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def kkt_errors(stdout):
    init = float(re.search(r"Initial KKT error = ([-+0-9.eE]+|nan|inf)", stdout).group(1))
    its = [float(x) for x in re.findall(r"KKT error after iteration \d+ = ([-+0-9.eE]+|nan|inf)", stdout)]
    return init, its


def test_anymal_ocp_benchmark_example_matches_oracle():
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_ocp_benchmark"), ANYMAL_URDF, "20"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 10 and "CPU time per update" in r.stdout
    # the same problem through the oracle (examples/anymal/ocp_benchmark.cpp:31-114 as data)
    model = anymal_model()
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", ANYMAL_Q_STANDING)
    cost.set("q_weight", np.full(nv, 10.0)).set("qf_weight", np.full(nv, 10.0))
    cost.set("v_weight", np.ones(nv)).set("vf_weight", np.ones(nv)).set("a_weight", np.full(nv, 0.01))
    for c in range(4):
        for k in range(3):
            cost.f_weight[c][k] = 0.001
            cost.f_ref[c][k] = 0.0
        cost.f_ref[c][2] = 70.0
    cons = capi.Constraints()
    import ctypes as C
    capi.lib().idocp_constraints_init(C.byref(cons))
    cons.friction_cone = 1                                  # FrictionCone(robot, 0.7), ocp_benchmark.cpp:76
    cons.mu = 0.7
    o = OracleOCP(model, cost, cons, 0.5, 20)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(nv)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(model))
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init_constraints(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)          # printed with 6 significant digits
    for k in range(10):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 2e-5 * max(1.0, ref), (k, its[k], ref)
    assert its[-1] < init


def test_iiwa14_unocp_benchmark_example_matches_oracle():
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "iiwa14_unocp_benchmark"), IIWA_URDF, "20"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) >= 10 and its[-1] < init


def test_iiwa14_unparnmpc_benchmark_example_matches_oracle():
    """examples/iiwa14_unparnmpc_benchmark.cpp = the reference's examples/iiwa14/unparnmpc_benchmark.cpp through the facade
    (idocp::UnParNMPCSolver): the printed KKT errors follow the oracle's iteration and reach its floor."""
    from helpers import OracleUnParNMPC, iiwa14_model, unocp_problem
    build_examples()
    out = os.path.join(ROOT, "gpurun_out", "unparnmpc_solution")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([os.path.join(ROOT, "examples", "iiwa14_unparnmpc_benchmark"), IIWA_URDF, out], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 100 and "feasible: yes" in r.stdout
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnParNMPC(m, cost, cons, 1.0, 20)
    q, v = np.full(m.nv, 2.0), np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init(0.0)
    assert abs(init - o.kkt_error(0.0, q, v)) < 1e-5 * init          # printed with 6 significant digits
    for k in range(3):
        assert o.update(0.0, q, v) == 0
        e = o.kkt_error(0.0, q, v)
        assert abs(its[k] - e) < 1e-5 * e, (k, its[k], e)
    assert its[-1] < 1e-6
    # saveSolution / printSolution (unparnmpc_solver.cpp:243-327): N lines of dimq / dimu numbers, the converged trajectory
    for _ in range(97):
        assert o.update(0.0, q, v) == 0
    Q, U = np.loadtxt(out + "_q.txt"), np.loadtxt(out + "_u.txt")
    assert Q.shape == (20, m.nv) and U.shape == (20, m.nv)
    assert np.abs(Q - o.get("q")).max() < 1e-4 and np.abs(U - o.get("u")).max() < 1e-2      # 6 significant digits in the file
    assert "v[19] = " in r.stdout


def test_anymal_trotting_example_matches_oracle():
    """examples/anymal_trotting.cpp = the reference's examples/anymal/anymal_trotting.cpp driver (contact sequence with a
    lift and two impulse events) through the facade: its printed KKT errors follow the oracle's hybrid SQP iteration."""
    from helpers import anymal_problem, trotting_sequence
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_trotting"), ANYMAL_URDF], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 25
    model = anymal_model()
    cost, cons = anymal_problem(model, trotting_ref=True)
    o = OracleOCP(model, cost, cons, 1.55, 30, max_num_impulse=3)
    trotting_sequence(o, model, 2)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(model.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init_constraints(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)          # printed with 6 significant digits
    for k in range(25):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 2e-5 * max(1.0, ref) + 1e-9, (k, its[k], ref)
    assert its[-1] < 1e-8


def test_anymal_running_example_matches_oracle():
    """examples/anymal_running.cpp = the reference's examples/anymal/anymal_running.cpp driver (TimeVaryingConfigurationSpaceCost,
    40 discrete events with flight phases, N = 240, T = 7) through the facade, 20 SQP iterations."""
    from helpers import ANYMAL_Q_RUNNING_START, running_problem, running_sequence
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_running"), ANYMAL_URDF, "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 20
    model = anymal_model()
    cost, cons = running_problem(model, 10)
    o = OracleOCP(model, cost, cons, 7.0, 240, max_num_impulse=26)
    assert running_sequence(o, model, 10) == 40
    q, v = ANYMAL_Q_RUNNING_START.copy(), np.zeros(model.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init_constraints(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)          # printed with 6 significant digits
    for k in range(20):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 5e-5 * max(1.0, ref), (k, its[k], ref)
    assert its[-1] < 0.05 * init


def test_anymal_jumping_example_matches_oracle():
    """examples/anymal_jumping.cpp = the reference's examples/anymal/anymal_jumping.cpp driver (three jumps: flight phases without
    contacts, landings as impulse stages with all twelve rows and twelve-row switching constraints; N = 100, T = 5) through the
    facade, 25 SQP iterations against the oracle on the same problem."""
    from helpers import jumping_problem, jumping_sequence
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_jumping"), ANYMAL_URDF, "25"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 25
    model = anymal_model()
    cost, cons = jumping_problem(model)
    o = OracleOCP(model, cost, cons, 5.0, 100, max_num_impulse=3)
    assert jumping_sequence(o, model, 3) == 6
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(model.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init_constraints(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)          # printed with 6 significant digits
    for k in range(25):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 5e-5 * max(1.0, ref), (k, its[k], ref)


def test_anymal_parnmpc_benchmark_example_matches_oracle():
    from helpers import OracleParNMPC
    import ctypes as C
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_parnmpc_benchmark"), ANYMAL_URDF, "20"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 20
    model = anymal_model()
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", ANYMAL_Q_STANDING)
    cost.set("q_weight", np.full(nv, 10.0)).set("qf_weight", np.full(nv, 10.0))
    cost.set("v_weight", np.ones(nv)).set("vf_weight", np.ones(nv)).set("a_weight", np.full(nv, 0.01))
    for c in range(4):
        for k in range(3):
            cost.f_weight[c][k] = 0.001
            cost.f_ref[c][k] = 0.0
        cost.f_ref[c][2] = 70.0
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    cons.friction_cone = 1                                  # FrictionCone(robot, 0.7), parnmpc_benchmark.cpp:76
    cons.mu = 0.7
    o = OracleParNMPC(model, cost, cons, 0.5, 20)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(nv)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(model))
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)
    for k in range(20):
        o.update(0.0, q, v)
        ref = o.kkt_error(0.0, q, v)
        assert abs(its[k] - ref) <= 2e-5 * max(1.0, ref) + 1e-9, (k, its[k], ref)


def test_anymal_trotting_parnmpc_example_matches_oracle():
    """examples/anymal_trotting_parnmpc.cpp = the reference's examples/anymal/anymal_trotting_parnmpc.cpp driver (lift + impulse
    events on a ParNMPC horizon) through the facade.  Full ParNMPC steps from this cold start do not contract (the reference runs
    200 iterations without a line search), so only the first iterations are compared, before the two trajectories of the same
    unstable iteration part."""
    from helpers import OracleParNMPC, anymal_problem
    build_examples()
    r = subprocess.run([os.path.join(ROOT, "examples", "anymal_trotting_parnmpc"), ANYMAL_URDF, "3"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    init, its = kkt_errors(r.stdout)
    assert len(its) == 3
    model = anymal_model()
    cost, cons = anymal_problem(model, trotting_ref=True)
    cons.linearized_friction_cone = 0
    cons.linearized_impulse_friction_cone = 0
    o = OracleParNMPC(model, cost, cons, 1.55, 60, max_num_impulse=3)
    pts = anymal_contact_points(model).copy()
    o.set_contact_status([1, 1, 1, 1], pts)
    o.push_back_contact_status([0, 1, 1, 0], pts, 0.5)
    pts[0, 0] += 0.075
    pts[3, 0] += 0.075
    o.push_back_contact_status([1, 0, 0, 1], pts, 1.0)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(model.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    o.init(0.0)
    ref_init = o.kkt_error(0.0, q, v)
    assert abs(init - ref_init) <= 1e-5 * max(1.0, ref_init)
    assert o.update(0.0, q, v) == 0
    ref = o.kkt_error(0.0, q, v)
    assert abs(its[0] - ref) <= 1e-4 * max(1.0, ref), (its[0], ref)
    assert np.isfinite(its).all()


def test_facade_surface_const_getters_copies_and_default_constructors(tmp_path):
    # ocp_solver.hpp:44,97 and its siblings: `getSolution(stage) const` (one device-to-host copy, idocp_*_get_split_solution) equals
    # the field getters; deep copies of live solvers continue identically; default-constructed solvers can be assigned
    exe = str(tmp_path / "facade_surface")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/facade_surface.cpp"),
                        "-L" + os.path.join(ROOT, "idocp_amd/lib"), "-lidocp_hip", "-Wl,-rpath," + os.path.join(ROOT, "idocp_amd/lib"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, IIWA_URDF, ANYMAL_URDF], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "fixed-base solvers: ok" in r.stdout and "shared cost function: ok" in r.stdout and "contact-path solvers on the fixed-base robot: ok" in r.stdout and "floating-base solver: ok" in r.stdout and "receding horizon through the facade: ok" in r.stdout
