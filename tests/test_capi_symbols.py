"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/idocp_hip.h declares; argument errors follow the documented codes.  No
compute call is made here."""
import ctypes as C
import os
import re

from idocp_amd import capi
from helpers import ROOT, iiwa14_model, unocp_problem


def declared_functions():
    text = open(os.path.join(ROOT, "include", "idocp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(idocp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = capi.lib()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_defaults():
    lib = capi.lib()
    assert b"gfx950" in lib.idocp_version()
    c = capi.Constraints()
    lib.idocp_constraints_init(C.byref(c))
    assert c.barrier == 1.0e-04 and c.fraction_to_boundary_rate == 0.995


def test_create_argument_errors():
    # UnOCPSolver's constructor checks (src/unocp/unocp_solver.cpp:33-47) surface as IDOCP_E_ARG
    lib = capi.lib()
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    h = C.c_void_p()
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), -1.0, 20, 1, 0, C.byref(h)) == -1
    assert b"T must be positive" in lib.idocp_last_error()
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 0, 1, 0, C.byref(h)) == -1
    assert b"N must be positive" in lib.idocp_last_error()
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 20, 0, 0, C.byref(h)) == -1


def test_no_cpu_fallback_without_gpu():
    # On a box without a GPU creating a solver must fail loudly (IDOCP_E_DEVICE), never fall back.
    lib = capi.lib()
    n = C.c_int()
    lib.idocp_device_count(C.byref(n))
    if n.value > 0:
        return
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    h = C.c_void_p()
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 20, 1, 0, C.byref(h)) == -3
    assert b"no CPU fallback" in lib.idocp_last_error()


def test_contact_positions_host_fk_matches_oracle():
    """idocp_model_contact_positions is host arithmetic (problem set-up), so it can be checked
    without a GPU: feet of ANYmal at q_standing and at a random configuration vs the oracle's
    frame kinematics (Robot::updateFrameKinematics + getContactPoints, robot.hxx:85-91, 262-283)."""
    import ctypes as C
    import numpy as np
    from helpers import ANYMAL_Q_STANDING, P, anymal_model, arr, oracle
    from idocp_amd import capi
    lib = capi.lib()
    model = anymal_model()
    olib = oracle()
    rng = np.random.default_rng(3)
    for trial in range(3):
        q = arr(ANYMAL_Q_STANDING).copy()
        if trial > 0:
            q[0:3] += rng.uniform(-0.3, 0.3, 3)
            quat = rng.normal(size=4)
            q[3:7] = quat / np.linalg.norm(quat)
            q[7:] += rng.uniform(-0.5, 0.5, 12)
        pts = np.zeros((model.ncontacts, 3))
        assert lib.idocp_model_contact_positions(C.byref(model), P(q), P(pts)) == 0
        nv, nc = model.nv, model.ncontacts
        z = np.zeros(nv)
        fp = np.zeros((nc, 3))
        tmp = [np.zeros(n) for n in (3 * nc, 3 * nc * nv, 3 * nc * nv, 3 * nc * nv)]
        fR, fv, fa = np.zeros((nc, 9)), np.zeros((nc, 6)), np.zeros((nc, 6))
        d4 = [np.zeros(nc * 6 * nv) for _ in range(4)]
        olib.oracle_contact_kinematics(C.byref(model), P(q), P(z), P(z), P(np.zeros((nc, 3))), C.c_double(0.05), P(tmp[0]), P(tmp[1]),
                                       P(tmp[2]), P(tmp[3]), P(fp), P(fR), P(fv), P(fa), P(d4[0]), P(d4[1]), P(d4[2]), P(d4[3]), None)
        assert np.abs(pts - fp).max() < 1e-13


def test_facade_object_semantics_without_gpu(tmp_path):
    # default construction, copy and move of EMPTY solvers (ocp_solver.hpp:44-74 and its three siblings) need no device
    import subprocess
    exe = str(tmp_path / "facade_surface")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/facade_surface.cpp"),
                        "-L" + os.path.join(ROOT, "idocp_amd/lib"), "-lidocp_hip", "-Wl,-rpath," + os.path.join(ROOT, "idocp_amd/lib"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "empty solvers: ok" in r.stdout, r.stdout + r.stderr


def test_acceleration_bound_errors():
    # JointAcceleration*Limit bounds are validated by create (finite; a_min < a_max where both are in use): IDOCP_E_ARG
    lib = capi.lib()
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    h = C.c_void_p()
    cons.joint_acceleration_lower_limit = 1
    cons.joint_acceleration_upper_limit = 1
    for r in range(m.nu):
        cons.a_min[r], cons.a_max[r] = -5.0, 5.0
    cons.a_max[3] = -6.0
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 20, 1, 0, C.byref(h)) == -1
    assert b"a_min must be smaller than a_max" in lib.idocp_last_error()
    cons.a_max[3] = float("inf")
    assert lib.idocp_unocp_create(C.byref(m), C.byref(cost), C.byref(cons), 1.0, 20, 1, 0, C.byref(h)) == -1
    assert b"must be finite" in lib.idocp_last_error()
    from helpers import anymal_model, anymal_problem
    ma = anymal_model()
    costa, consa = anymal_problem(ma)
    consa.joint_acceleration_upper_limit = 1
    consa.a_max[0] = float("nan")
    assert lib.idocp_ocp_create(C.byref(ma), C.byref(costa), C.byref(consa), 1.0, 20, 1, 0, C.byref(h)) == -1
    assert b"must be finite" in lib.idocp_last_error()
