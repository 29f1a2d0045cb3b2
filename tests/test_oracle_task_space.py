"""TaskSpace3DCost / TaskSpace6DCost (+ TimeVarying variants) in the CPU restatement of UnOCPSolver: the terms against
finite differences and the identities of the reference's formulas, and the SQP of examples/iiwa14/task_space_ocp.cpp (CPU only)."""
import numpy as np
import pytest

from helpers import OracleUnOCP, iiwa14_model
from idocp_amd.workloads import task_circle_refs, task_space_problem

Q0 = np.array([0, np.pi / 2, 0, np.pi / 2, 0, np.pi / 2, 0.0])       # task_space_ocp.cpp:86


def make(dim, N=20, T=1.0, time_varying=False, weight=1000.0):
    m = iiwa14_model()
    cost, cons = task_space_problem(m, dim=dim, time_varying=time_varying, weight=weight)
    o = OracleUnOCP(m, cost, cons, T, N)
    o.set_solution("q", Q0)
    o.set_solution("v", np.zeros(7))
    return m, cost, o


@pytest.mark.parametrize("dim", [3, 6])
def test_gradient_is_the_derivative_of_the_cost_and_the_hessian_is_gauss_newton(dim):
    # task_space_3d_cost.cpp:86-157 / time_varying_task_space_6d_cost.cpp:108-192: lq += JJ^T W diff, Qqq += JJ^T W JJ with
    # JJ = R_frame J_lin (3D) or Jlog6(M_ref^-1 M) J (6D) -- i.e. JJ = d diff / dq
    m, cost, o = make(dim)
    rng = np.random.default_rng(2)
    for trial in range(4):
        q = Q0 + 0.3 * rng.uniform(-1, 1, 7)
        c, g, H = o.task_terms(3, q)
        assert c > 0 and np.allclose(H, H.T, atol=1e-9 * np.abs(H).max()) and np.linalg.eigvalsh(H).min() > -1e-8 * np.abs(H).max()
        h = 1e-6
        gfd = np.zeros(7)
        for k in range(7):
            e = np.zeros(7); e[k] = h
            gfd[k] = (o.task_terms(3, q + e)[0] - o.task_terms(3, q - e)[0]) / (2 * h)
        assert np.allclose(g, gfd, rtol=1e-6, atol=1e-6 * np.abs(g).max())
        # Gauss-Newton: H = Jr^T W Jr with Jr = d diff / dq; with w = 1000 on every component H = 1000 Jr^T Jr and
        # g = 1000 Jr^T diff, so the directional derivative of g along d is H d + (second-order term that vanishes with diff)
        # -- checked where diff -> 0: at the reference configuration the Hessian is the derivative of the gradient
    # configuration that realises the reference: gradient 0, and there the FD Jacobian of the gradient equals H
    q = Q0.copy()
    for _ in range(30):
        c, g, H = o.task_terms(3, q)
        q = q - np.linalg.solve(H + 1e-6 * np.eye(7), g)
    c, g, H = o.task_terms(3, q)
    assert c < 1e-10
    Hfd = np.zeros((7, 7))
    for k in range(7):
        e = np.zeros(7); e[k] = 1e-6
        Hfd[:, k] = (o.task_terms(3, q + e)[1] - o.task_terms(3, q - e)[1]) / 2e-6
    assert np.allclose(H, Hfd, rtol=1e-4, atol=1e-4 * np.abs(H).max())


def test_weights_of_the_6d_cost_follow_the_reference_layout():
    # time_varying_task_space_6d_cost.cpp:33-38: q_6d_weight = [rotation_weight; position_weight] multiplies log6 = [linear; angular],
    # so the vector handed over as "rotation_weight" prices the translational part.  At a configuration whose frame pose equals the
    # reference, shifting the reference POSITION by d gives log6 = [-R_ref^T d; 0]: cost |d|^2 / 2 under the head weights, 0 under the tail
    m = iiwa14_model()
    cost, cons = task_space_problem(m, dim=6)
    o = OracleUnOCP(m, cost, cons, 1.0, 20)
    q = Q0.copy()
    for _ in range(30):
        c, g, H = o.task_terms(0, q)
        q = q - np.linalg.solve(H + 1e-6 * np.eye(7), g)
    assert o.task_terms(0, q)[0] < 1e-12
    d = np.array([0.03, -0.02, 0.05])
    out = {}
    for name, w in (("head", [1, 1, 1, 0, 0, 0]), ("tail", [0, 0, 0, 1, 1, 1])):
        cost2, _ = task_space_problem(m, dim=6)
        for k in range(6):
            cost2.task_weight[k] = float(w[k])
        for k in range(3):
            cost2.task_ref[9 + k] += d[k]
        out[name] = OracleUnOCP(m, cost2, cons, 1.0, 20).task_terms(0, q)[0]
    assert abs(out["head"] - 0.5 * d @ d) < 1e-9 and abs(out["tail"]) < 1e-9, out


def test_sqp_of_the_task_space_example_converges():
    # examples/iiwa14/task_space_ocp.cpp:54-99 (T = 6, N = 120 there; a shorter horizon here), time-varying circular reference
    m, cost, o = make(6, N=30, T=1.5, time_varying=True)
    o.set_task_refs(task_circle_refs(0.0, 1.5 / 30, 30))
    e0 = o.kkt_error(0.0, Q0, np.zeros(7))
    errs = [e0]
    for _ in range(40):
        assert o.update(0.0, Q0, np.zeros(7)) == 0
        errs.append(o.kkt_error(0.0, Q0, np.zeros(7)))
    assert np.isfinite(errs).all() and errs[-1] < 1e-6 * e0, errs[-5:]
    # the end effector follows the circle at the end of the horizon
    qN = o.solution("q")[-1]
    cN = o.task_terms(30, qN)[0]
    assert cN < 1.0          # 1000/2 |log6|^2 < 1: the pose error is below 5 cm / 3 deg


def test_facade_components_export_the_reference_layout(tmp_path):
    # include/idocp/cost/task_space_cost.hpp: what TaskSpace3DCost / TaskSpace6DCost / TimeVarying* write into the flat cost block
    # (weights in the reference's storage order [rotation; position], refs row-major + position, one pose per stage from the user's
    # TimeVaryingTaskSpace*RefBase), and that pushing a ConfigurationSpaceCost before or after keeps them.  No GPU involved.
    import os
    import subprocess
    from helpers import ROOT, IIWA_URDF
    exe = str(tmp_path / "task_cost_fields")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/task_cost_fields.cpp"),
                    "-L" + os.path.join(ROOT, "idocp_amd/lib"), "-lidocp_hip", "-Wl,-rpath," + os.path.join(ROOT, "idocp_amd/lib"), "-o", exe],
                   check=True, capture_output=True)
    out = subprocess.run([exe, IIWA_URDF], check=True, capture_output=True, text=True).stdout
    rec = {}
    for line in out.splitlines():
        tok = line.split()
        d, key = {"name": tok[0]}, None
        for t in tok[1:]:
            try:
                d.setdefault(key, []).append(float(t))
            except ValueError:
                key = t
                d[key] = []
        rec[tok[0]] = d
    m = iiwa14_model()
    ref_cost, _ = task_space_problem(m, dim=6)
    frame = list(ref_cost.task_frame_R) + list(ref_cost.task_frame_p)
    for name, dim, tv in (("3d", 3, 0), ("6d", 6, 0), ("tv3d", 3, 1), ("tv6d", 6, 1)):
        r = rec[name]
        assert r["dim"] == [dim] and r["joint"] == [ref_cost.task_joint] and r["tv"] == [tv] and r["frame"] == frame, name
    assert rec["3d"]["weight"] == [1, 2, 3, 0, 0, 0] and rec["3d"]["weightf"] == [4, 5, 6, 0, 0, 0]
    assert rec["3d"]["ref"][9:] == [0.1, 0.2, 0.3] and rec["3d"]["vweight"] == [0.5] and rec["3d"]["refs"] == []
    # set_q_6d_weight(position, rotation) -> [rotation; position]  (task_space_6d_cost.cpp:48-60)
    assert rec["6d"]["weight"] == [7, 8, 9, 1, 2, 3] and rec["6d"]["weightf"] == [10, 11, 12, 4, 5, 6] and rec["6d"]["vweight"] == [0.5]
    assert rec["6d"]["ref"] == [0, -1, 0, 1, 0, 0, 0, 0, 1, 0.1, 0.2, 0.3]
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    assert rec["tv3d"]["refs"] == eye + [1.5, -0.25, 0.75] + eye + [2.0, -0.375, 0.75] + eye + [2.5, -0.5, 0.75]
    R = [0, 0, 1, 0, 1, 0, -1, 0, 0]
    assert rec["tv6d"]["weight"] == [100] * 3 + [1000] * 3 and rec["tv6d"]["weightf"] == [1] * 3 + [10] * 3
    assert np.allclose(rec["tv6d"]["refs"], R + [0.5, 0.1, 0.7] + R + [0.5, 0.15, 0.7] + R + [0.5, 0.2, 0.7], atol=1e-15)


def test_task_space_cost_on_the_floating_base_oracle_converges():
    """TaskSpace3DCost on a foot-less frame of ANYmal (the thigh of RH) in OCPSolver's oracle: stage, terminal weights; the SQP converges
    and the frame moves towards its reference."""
    import ctypes as C
    from idocp_amd import capi
    from idocp_amd.workloads import ANYMAL_URDF
    from helpers import ANYMAL_Q_STANDING, OracleOCP, anymal_contact_points, anymal_model, anymal_problem
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    lib = capi.lib()
    fid = lib.idocp_model_frame_id(ANYMAL_URDF.encode(), b"RH_THIGH")
    joint = C.c_int()
    R, p = (C.c_double * 9)(), (C.c_double * 3)()
    assert lib.idocp_model_frame_placement(ANYMAL_URDF.encode(), fid, C.byref(joint), R, p) == 0
    cost.task_dim, cost.task_joint = 3, joint.value
    for k in range(9):
        cost.task_frame_R[k] = R[k]
        cost.task_ref[k] = 1.0 if k % 4 == 0 else 0.0
    for k in range(3):
        cost.task_frame_p[k] = p[k]
        cost.task_weight[k] = 200.0
        cost.task_weightf[k] = 200.0
    results = []
    for shift in (0.0, 0.03):
        for k, x in enumerate((-0.277 , -0.116 + shift, 0.4792)):       # near the hip of RH in the standing pose, then 3 cm to the side
            cost.task_ref[9 + k] = x
        o = OracleOCP(m, cost, cons, 0.5, 20)
        q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
        o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
        o.set_solution("q", q); o.set_solution("v", v)
        o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        o.init_constraints(0.0)
        e0 = o.kkt_error(0.0, q, v)
        for _ in range(40):
            assert o.update(0.0, q, v) == 0
        assert o.kkt_error(0.0, q, v) < 1e-8 * max(1.0, e0)
        results.append(o.get("q")[-1].copy())
    # the base follows the reference: it ends further in +y when the reference is shifted in +y
    assert results[1][1] - results[0][1] > 0.005
