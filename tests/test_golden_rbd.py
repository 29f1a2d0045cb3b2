"""Oracle + URDF reader against the committed golden vectors (CPU only).

tests/golden/rbd_*.json come from tests/golden/gen_golden_rbd.py: an independent
numpy body-frame RNEA with complex-step derivatives and its own URDF walk.  The
oracle restates pinocchio's world-frame analytic algorithm, so agreement pins
both the derivation and the model conventions (joint order, frame ids, inertia
merging of fixed joints)."""
import ctypes as C

import numpy as np
import pytest

from idocp_amd import capi
from helpers import ANYMAL_CONTACT_FRAMES, ANYMAL_URDF, IIWA_URDF, P, arr, load_golden, oracle, rel_err

ROBOTS = [("iiwa14", IIWA_URDF, ()), ("anymal", ANYMAL_URDF, ANYMAL_CONTACT_FRAMES)]


@pytest.mark.parametrize("robot,urdf,frames", ROBOTS)
def test_urdf_reader_matches_golden_model(robot, urdf, frames):
    g = load_golden(robot)["model"]
    m = capi.model_from_urdf(urdf, frames)
    assert (m.njoints, m.nq, m.nv, m.has_floating_base) == (g["njoints"], g["nq"], g["nv"], g["floating"])
    assert m.nu == m.nv - (6 if g["floating"] else 0)
    for i in range(m.njoints):
        assert m.parent[i] == g["parent"][i] and m.jtype[i] == g["jtype"][i]
        assert m.idx_q[i] == g["idx_q"][i] and m.idx_v[i] == g["idx_v"][i]
        np.testing.assert_allclose(m.axis[i][:], g["axis"][i], atol=1e-15)
        np.testing.assert_allclose(m.plc_R[i][:], g["plc_R"][i], atol=1e-15)
        np.testing.assert_allclose(m.plc_p[i][:], g["plc_p"][i], atol=1e-15)
        np.testing.assert_allclose(m.mass[i], g["mass"][i], rtol=1e-15)
        np.testing.assert_allclose(m.com[i][:], g["com"][i], atol=1e-15)
        np.testing.assert_allclose(m.inertia[i][:], g["inertia"][i], atol=1e-15)
    for k in range(m.nu):
        assert m.q_min[k] == g["q_min"][k] and m.q_max[k] == g["q_max"][k]
        assert m.v_max[k] == g["v_max"][k] and m.u_max[k] == g["u_max"][k]
    for c, gc in enumerate(g["contacts"]):
        assert m.contact_frame_id[c] == gc["frame"] and m.contact_joint[c] == gc["joint"]
        np.testing.assert_allclose(m.contact_R[c][:], gc["R"], atol=1e-15)
        np.testing.assert_allclose(m.contact_p[c][:], gc["p"], atol=1e-15)


def test_frame_ids_match_reference_examples():
    # examples/anymal/anymal_trotting.cpp:30 uses frames {14,24,34,44} = LF,LH,RF,RH feet;
    # the reference tests use frame 18 of iiwa14
    lib = capi.lib()
    for name, fid in (("LF_FOOT", 14), ("LH_FOOT", 24), ("RF_FOOT", 34), ("RH_FOOT", 44)):
        assert lib.idocp_model_frame_id(ANYMAL_URDF.encode(), name.encode()) == fid
    assert lib.idocp_model_frame_id(IIWA_URDF.encode(), b"iiwa_link_7") == 18
    assert lib.idocp_model_frame_id(IIWA_URDF.encode(), b"no_such_frame") == -1


def test_urdf_errors():
    m = capi.Model()
    lib = capi.lib()
    assert lib.idocp_model_from_urdf(b"/nonexistent.urdf", None, 0, C.byref(m)) == -2
    bad = (C.c_int * 1)(9999)
    assert lib.idocp_model_from_urdf(IIWA_URDF.encode(), bad, 1, C.byref(m)) == -1


@pytest.mark.parametrize("robot,urdf,frames", ROBOTS)
def test_oracle_rnea_and_derivatives_match_golden(robot, urdf, frames):
    g = load_golden(robot)
    m = capi.model_from_urdf(urdf, frames)
    nv = m.nv
    ol = oracle()
    for s in g["samples"]:
        q, v, a = arr(s["q"]), arr(s["v"]), arr(s["a"])
        f = arr(s["f"]).reshape(-1) if "f" in s else None
        fp = P(f) if f is not None else None
        tau, dq, dv, da = np.zeros(nv), np.zeros((nv, nv)), np.zeros((nv, nv)), np.zeros((nv, nv))
        ol.oracle_rnea(C.byref(m), P(q), P(v), P(a), fp, 1, P(tau))
        ol.oracle_rnea_derivatives(C.byref(m), P(q), P(v), P(a), fp, 1, P(dq), P(dv), P(da))
        assert rel_err(tau, s["tau"]) < 1e-13
        assert rel_err(dq.T, s["dtau_dq"]) < 1e-12
        assert rel_err(dv.T, s["dtau_dv"]) < 1e-12
        assert rel_err(da.T, s["dtau_da"]) < 1e-12
        # identities the reference's tests assert (test/robot/robot_test.cpp:535-567): M symmetric PD
        M = da.T
        assert np.abs(M - M.T).max() < 1e-13 and np.linalg.eigvalsh(M).min() > 0
        if "tau_impulse" in s:      # impulse model: zero gravity, v = 0 (robot.hxx:505-540)
            z = np.zeros(nv)
            ol.oracle_rnea(C.byref(m), P(q), P(z), P(a), fp, 0, P(tau))
            ol.oracle_rnea_derivatives(C.byref(m), P(q), P(z), P(a), fp, 0, P(dq), P(dv), P(da))
            assert rel_err(tau, s["tau_impulse"]) < 1e-13
            assert rel_err(dq.T, s["dimp_dq"]) < 1e-12
            assert rel_err(da.T, s["dimp_ddv"]) < 1e-12


def test_oracle_contact_kinematics_baumgarte_mjtjinv_match_golden():
    """Frame kinematics and their derivatives (pinocchio getFrame*Derivatives, LOCAL),
    the Baumgarte residual/derivatives as PointContact assembles them, and MJtJinv."""
    import json
    import os
    from helpers import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "contact_anymal.json")))
    m = capi.model_from_urdf(ANYMAL_URDF, ANYMAL_CONTACT_FRAMES)
    nv, nc = m.nv, m.ncontacts
    ol = oracle()
    for s in g["samples"]:
        q, v, a, cp = arr(s["q"]), arr(s["v"]), arr(s["a"]), arr(s["contact_points"])
        C = np.zeros(3 * nc)
        dq, dv, da = (np.zeros((nv, 3 * nc)) for _ in range(3))          # col-major (3nc x nv)
        fp, fR, fv, fa = np.zeros((nc, 3)), np.zeros((nc, 9)), np.zeros((nc, 6)), np.zeros((nc, 6))
        vdq, adq, adv, ada = (np.zeros((nc, nv, 6)) for _ in range(4))    # col-major (6 x nv) per contact
        K = np.zeros((nv + 3 * nc, nv + 3 * nc))
        ol.oracle_contact_kinematics(C_byref(m), P(q), P(v), P(a), P(cp), C_double(s["time_step"]), P(C), P(dq), P(dv), P(da),
                                     P(fp), P(fR), P(fv), P(fa), P(vdq), P(adq), P(adv), P(ada), P(K))
        assert rel_err(fp, s["frame_p"]) < 1e-13 and rel_err(fR, s["frame_R"]) < 1e-13
        assert rel_err(fv, s["frame_v"]) < 1e-12 and rel_err(fa, s["frame_a"]) < 1e-12
        assert rel_err(vdq.transpose(0, 2, 1), s["v_partial_dq"]) < 1e-12
        assert rel_err(adq.transpose(0, 2, 1), s["a_partial_dq"]) < 1e-12
        assert rel_err(adv.transpose(0, 2, 1), s["a_partial_dv"]) < 1e-12
        assert rel_err(ada.transpose(0, 2, 1), s["a_partial_da"]) < 1e-12
        assert rel_err(C, s["C"]) < 1e-12
        assert rel_err(dq.T, s["dCdq"]) < 1e-12 and rel_err(dv.T, s["dCdv"]) < 1e-12 and rel_err(da.T, s["dCda"]) < 1e-12
        # MJtJinv: golden = dense inverse of [M J^T; J 0]; also the identity test/robot/robot_test.cpp:535-567 asserts
        assert rel_err(K.T, s["MJtJinv"]) < 1e-9
        Mm = np.linalg.inv(np.array(s["MJtJinv"]))
        assert np.abs(K.T @ Mm - np.eye(nv + 3 * nc)).max() < 1e-9


def test_oracle_lie_group_operations_match_golden():
    """integrate / difference / dDifference (ARG0, ARG1) of SE(3) x R^12 (robot.hxx:23-163)."""
    import json
    import os
    from helpers import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "contact_anymal.json")))
    m = capi.model_from_urdf(ANYMAL_URDF, ANYMAL_CONTACT_FRAMES)
    nv = m.nv
    ol = oracle()
    for s in g["samples"]:
        q, q1, dv = arr(s["q"]), arr(s["q1"]), arr(s["dv"])
        qi, diff, J0, J1 = np.zeros(m.nq), np.zeros(nv), np.zeros((nv, nv)), np.zeros((nv, nv))
        ol.oracle_lie_ops(C_byref(m), P(q), P(q1), P(dv), P(qi), P(diff), P(J0), P(J1))
        gq = arr(s["q_plus_dv"])
        if np.dot(gq[3:7], qi[3:7]) < 0:
            gq[3:7] *= -1                      # quaternion double cover
        assert rel_err(qi, gq) < 1e-12
        assert rel_err(diff, s["q1_minus_q"]) < 1e-12
        # golden Jacobians are Richardson-extrapolated differences (~1e-9 accurate)
        assert rel_err(J0.T, s["dDiff_arg0"]) < 1e-7 and rel_err(J1.T, s["dDiff_arg1"]) < 1e-7
        # block-triangular structure the reference's 6x6 inverse relies on (robot.hxx:151-163)
        assert np.abs(J1.T[3:6, 0:3]).max() < 1e-14 and np.abs(J0.T[3:6, 0:3]).max() < 1e-14


def C_byref(x):
    return C.byref(x)


def C_double(x):
    return C.c_double(x)
