"""Robot::integrateConfiguration / subtractConfiguration / normalizeConfiguration on the host (idocp_model_*_configuration, include/idocp_hip.h): what an
MPC loop does between two solver calls.  No GPU involved: the library's host code against the oracle's Robot (oracle/rbd.cpp, itself pinned by
tests/test_golden_rbd.py), and the identities the reference's robot_test.cpp checks (integrate then subtract gives the step back)."""
import ctypes as C

import numpy as np

from helpers import ANYMAL_Q_STANDING, P, anymal_model, arr, iiwa14_model, oracle
from idocp_amd import capi


def oracle_lie(model, q, q1, dv):
    lib = oracle()
    nq, nv = model.nq, model.nv
    q_int, diff, J0, J1 = np.zeros(nq), np.zeros(nv), np.zeros(nv * nv), np.zeros(nv * nv)
    assert lib.oracle_lie_ops(C.byref(model), P(arr(q)), P(arr(q1)), P(arr(dv)), P(q_int), P(diff), P(J0), P(J1)) == 0
    return q_int, diff


def random_configuration(model, rng):
    q = rng.uniform(-1, 1, model.nq)
    if model.has_floating_base:
        q[3:7] /= np.linalg.norm(q[3:7])
    return q


def test_integrate_and_subtract_match_the_oracle_and_invert_each_other():
    lib = capi.lib()
    rng = np.random.default_rng(7)
    for model in (anymal_model(), iiwa14_model()):
        nq, nv = model.nq, model.nv
        for trial in range(20):
            q, q1 = random_configuration(model, rng), random_configuration(model, rng)
            dv = rng.uniform(-1, 1, nv) * (1e-9 if trial == 0 else (3.0 if trial == 1 else 0.7))      # tiny and large rotations too
            length = [1.0, 0.05, -0.3][trial % 3]
            ref_int, ref_diff = oracle_lie(model, q, q1, length * dv)
            out, diff = np.zeros(nq), np.zeros(nv)
            capi.check(lib.idocp_model_integrate_configuration(C.byref(model), P(arr(q)), P(arr(dv)), length, P(out)), "integrate")
            capi.check(lib.idocp_model_subtract_configuration(C.byref(model), P(arr(q1)), P(arr(q)), P(diff)), "subtract")
            assert np.abs(out - ref_int).max() <= 1e-13, (trial, np.abs(out - ref_int).max())
            assert np.abs(diff - ref_diff).max() <= 1e-12, (trial, np.abs(diff - ref_diff).max())
            if model.has_floating_base:
                assert abs(np.linalg.norm(out[3:7]) - 1.0) <= 1e-14
            # (q (+) length dv) (-) q = length dv, while the rotation stays inside the injectivity radius of log
            if np.linalg.norm(length * dv[3:6] if model.has_floating_base else 0.0) < 3.0:
                back = np.zeros(nv)
                capi.check(lib.idocp_model_subtract_configuration(C.byref(model), P(out), P(arr(q)), P(back)), "subtract")
                assert np.abs(back - length * dv).max() <= 1e-11 * max(1.0, np.abs(dv).max()), trial


def test_normalize_configuration():
    lib = capi.lib()
    m = anymal_model()
    q = ANYMAL_Q_STANDING.copy()
    q[3:7] = [0.2, -0.4, 0.1, 1.3]
    want = q.copy()
    want[3:7] /= np.linalg.norm(want[3:7])
    capi.check(lib.idocp_model_normalize_configuration(C.byref(m), P(q)), "normalize")
    assert np.array_equal(q[:3], want[:3]) and np.array_equal(q[7:], want[7:]) and np.abs(q[3:7] - want[3:7]).max() <= 1e-16
    q[3:7] = 0.0
    assert lib.idocp_model_normalize_configuration(C.byref(m), P(q)) != 0 and b"zero quaternion" in lib.idocp_last_error()
    mi = iiwa14_model()
    qi = np.linspace(-1, 1, mi.nq)
    keep = qi.copy()
    capi.check(lib.idocp_model_normalize_configuration(C.byref(mi), P(qi)), "normalize")
    assert np.array_equal(qi, keep)
    assert lib.idocp_model_integrate_configuration(C.byref(mi), None, P(qi), 1.0, P(qi)) != 0


def test_frame_world_placement_matches_the_independent_frame_kinematics():
    """Robot::framePosition / frameRotation / framePlacement of the facade: idocp_model_frame_placement (frame id -> joint + local placement) followed by
    idocp_model_frame_world_placement, against the forward kinematics of tests/golden/gen_golden_rbd.py (its own URDF walk and frame numbering) -- EVERY
    frame of both URDFs that sits on a joint; the ones fixed to the world have no joint to report and are refused."""
    import sys
    from helpers import ANYMAL_CONTACT_FRAMES, ANYMAL_URDF, GOLDEN, IIWA_URDF
    sys.path.insert(0, GOLDEN)
    import gen_golden_rbd as RBD
    lib = capi.lib()
    rng = np.random.default_rng(77)
    for urdf, contact_frames in ((ANYMAL_URDF, ANYMAL_CONTACT_FRAMES), (IIWA_URDF, ())):
        m = capi.model_from_urdf(urdf, contact_frames)
        every = RBD.load_model(urdf)["frames"]
        frames = [fid for fid, fr in enumerate(every) if fr[1] >= 0]
        assert len(frames) >= 15 and set(contact_frames) <= set(frames)
        M = RBD.load_model(urdf, frames)                   # (the generator treats the frames asked for as its "contacts")
        joint = C.c_int()
        Rl, pl, Rw, pw = np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(3)
        for fid, fr in enumerate(every):
            if fr[1] < 0:
                assert lib.idocp_model_frame_placement(urdf.encode(), fid, C.byref(joint), P(Rl), P(pl)) != 0
        for _ in range(2):
            q = RBD.random_q(M, rng)
            want = RBD.frame_kinematics(M, q, np.zeros(M["nv"]), np.zeros(M["nv"]))
            for fid, w in zip(frames, want):
                assert lib.idocp_model_frame_placement(urdf.encode(), fid, C.byref(joint), P(Rl), P(pl)) == 0
                assert lib.idocp_model_frame_world_placement(C.byref(m), P(arr(q)), joint.value, P(Rl), P(pl), P(Rw), P(pw)) == 0
                assert np.abs(pw - w["p"]).max() < 1e-13 and np.abs(Rw.reshape(3, 3) - w["R"]).max() < 1e-13, (urdf, fid)
    assert lib.idocp_model_frame_world_placement(C.byref(m), P(arr(q)), 99, P(Rl), P(pl), P(Rw), P(pw)) != 0


def test_host_lie_operations_match_the_independent_vectors():
    """The product's host integrate / subtract against tests/golden/contact_anymal.json (gen_golden_rbd.py: SE(3) exponential and logarithm written out on
    their own) -- no restatement in between."""
    import json
    import os
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "contact_anymal.json")) as fh:
        gold = json.load(fh)
    m = anymal_model()
    lib = capi.lib()
    for s in gold["samples"]:
        q, q1, dv = arr(s["q"]), arr(s["q1"]), arr(s["dv"])
        qi, diff = np.zeros(m.nq), np.zeros(m.nv)
        assert lib.idocp_model_integrate_configuration(C.byref(m), P(q), P(dv), 1.0, P(qi)) == 0
        assert lib.idocp_model_subtract_configuration(C.byref(m), P(q1), P(q), P(diff)) == 0
        want = arr(s["q_plus_dv"])
        if np.dot(want[3:7], qi[3:7]) < 0:
            want[3:7] *= -1                    # quaternion double cover
        assert np.abs(qi - want).max() < 1e-13 and np.abs(diff - arr(s["q1_minus_q"])).max() < 1e-12
