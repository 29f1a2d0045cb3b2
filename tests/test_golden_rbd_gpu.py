"""The rigid-body layer of the CONTACT-PATH kernels against the independent vectors of tests/golden (gen_golden_rbd.py: body-frame recursive Newton-Euler with
complex-step derivatives, its own frame kinematics and Baumgarte assembly, a dense inverse of [M J^T; J 0]) -- the kernels themselves, not the oracle, held to
the answer the oracle is pinned by.  The condensation kernel keeps MJtJinv, MJtJinv [dID; dC]/d(q, v) and MJtJinv [ID; C] of a stage
(ContactDynamicsData, contact_dynamics_data.hxx:8-29; read with idocp_ocp_get_contact_dynamics); one inverse gives M and J back, two products the derivatives
and the residuals: rows a1 - a5, a8 of SURVEY 8 as the nominal sweeps, the tangent items and the block-arrow inverse of ocp_condense_kernel computed them."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, HipOCP, P, anymal_model, anymal_problem, arr, rel_err
from idocp_amd import capi

pytestmark = pytest.mark.gpu
NV, NF = 18, 12


def contact_dynamics_of_stage_0(g, instance):
    n = NV + NF
    MJ, MJD, MJIDC = np.zeros(n * n), np.zeros(n * 2 * NV), np.zeros(n)
    dimf = g.lib.idocp_ocp_get_contact_dynamics(g.h, instance, 0, P(MJ), P(MJD), P(MJIDC))
    assert dimf == NF, (dimf, capi.lib().idocp_last_error())
    MJ, MJD = MJ.reshape(n, n).T, MJD.reshape(2 * NV, n).T      # column-major -> [row, column]
    K = np.linalg.inv(MJ)                                       # [M J^T; J 0]
    return MJ, K, K @ MJD, K @ MJIDC


def linearise(sample, f=None, time_step=0.05, contact_points=None):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    g = HipOCP(m, cost, cons, 2 * time_step, 2, batch=2)        # (the Baumgarte time step is T / N: hybrid_container.hpp:186-188)
    q, v, a = arr(sample["q"]), arr(sample["v"]), arr(sample["a"])
    pts = arr(contact_points) if contact_points is not None else np.zeros((4, 3))
    g.set_contact_status([1, 1, 1, 1], pts)
    g.set_solution("q", q)
    g.set_solution("v", v)
    g.set_solution("a", a)
    g.set_solution("u", np.zeros(12))
    ff = arr(f).reshape(1, 12) if f is not None else np.zeros((1, 12))
    capi.check(g.lib.idocp_ocp_set_solution_stages(g.h, b"f", 2, P(arr(np.repeat(ff, 2, axis=0)))), "set f")
    g.init_constraints(0.0)
    assert g.update(0.0, q, v) == 0, capi.lib().idocp_last_error()
    return g


def test_inverse_dynamics_and_its_derivatives_against_the_independent_vectors():
    """rbd_anymal.json: tau = ID(q, v, a, f) with four contact forces, d tau / d (q, v, a) by complex step (rows a1 - a3)."""
    with open(os.path.join(GOLDEN, "rbd_anymal.json")) as fh:
        gold = json.load(fh)
    worst = 0.0
    for s in gold["samples"]:
        g = linearise(s, f=s["f"])
        for inst in (0, 1):
            _, K, dIDC, IDC = contact_dynamics_of_stage_0(g, inst)
            errs = (rel_err(IDC[:NV], s["tau"]),                                    # u = 0: the residual of the inverse dynamics is tau itself
                    rel_err(dIDC[:NV, :NV], s["dtau_dq"]), rel_err(dIDC[:NV, NV:], s["dtau_dv"]), rel_err(K[:NV, :NV], s["dtau_da"]))
            worst = max(worst, *errs)
            assert max(errs) < 1e-11, errs
    print("ID, dID/dq, dID/dv, M against the independent vectors: worst %.2e" % worst)


def test_baumgarte_terms_and_mjtjinv_against_the_independent_vectors():
    """contact_anymal.json: the Baumgarte residual C and dC / d (q, v, a) as PointContact assembles them (rows a4, a5), MJtJinv as a dense inverse (a8)."""
    with open(os.path.join(GOLDEN, "contact_anymal.json")) as fh:
        gold = json.load(fh)
    worst = 0.0
    for s in gold["samples"]:
        g = linearise(s, time_step=s["time_step"], contact_points=s["contact_points"])
        for inst in (0, 1):
            MJ, K, dIDC, IDC = contact_dynamics_of_stage_0(g, inst)
            errs = (rel_err(MJ, s["MJtJinv"]), rel_err(IDC[NV:], s["C"]), rel_err(dIDC[NV:, :NV], s["dCdq"]), rel_err(dIDC[NV:, NV:], s["dCdv"]),
                    rel_err(K[NV:, :NV], s["dCda"]), float(np.abs(K[NV:, NV:]).max()))
            worst = max(worst, *errs)
            assert max(errs) < 1e-11, errs
    print("MJtJinv, C, dC/dq, dC/dv, J against the independent vectors: worst %.2e" % worst)


def test_impulse_dynamics_against_the_independent_vectors():
    """rbd_anymal.json `tau_impulse`, `dimp_dq`, `dimp_ddv`: Robot::RNEAImpulse(+Derivatives) (robot.hxx:505-540) -- zero gravity, v = 0, a = dv, the impulse
    forces on all four feet -- as the impulse instantiation of the condensation kernel computes them on the impulse stage of a landing out of a flight phase."""
    with open(os.path.join(GOLDEN, "rbd_anymal.json")) as fh:
        gold = json.load(fh)
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    worst = 0.0
    for s in gold["samples"]:
        g = HipOCP(m, cost, cons, 0.2, 4, batch=2, max_num_impulse=1)
        q, v, a = arr(s["q"]), arr(s["v"]), arr(s["a"])
        pts = np.zeros((4, 3))
        g.set_contact_status([0, 0, 0, 0], pts)
        g.push_back_contact_status([1, 1, 1, 1], pts, 0.12)
        g.set_solution("q", q)
        g.set_solution("v", v)
        g.set_solution("a", a)                                # (dv of the impulse stage: ocp_solver.cpp:116-123)
        g.set_solution("u", np.zeros(12))
        g.init_constraints(0.0)
        chain = g.chain(0.0)
        pos = [c["kind"] for c in chain].index("impulse")
        M = len(chain)
        f = np.zeros((M, 12))
        f[:] = arr(s["f"]).reshape(-1)
        capi.check(g.lib.idocp_ocp_set_solution_chain(g.h, b"f", M, P(arr(f))), "set f along the chain")
        assert g.update(0.0, q, v) == 0, capi.lib().idocp_last_error()
        n = NV + NF
        for inst in (0, 1):
            MJ, MJD, MJIDC = np.zeros(n * n), np.zeros(n * 2 * NV), np.zeros(n)
            assert g.lib.idocp_ocp_get_contact_dynamics_chain(g.h, inst, pos, P(MJ), P(MJD), P(MJIDC)) == NF
            K = np.linalg.inv(MJ.reshape(n, n).T)
            dImD, ImD = K @ MJD.reshape(2 * NV, n).T, K @ MJIDC
            errs = (rel_err(ImD[:NV], s["tau_impulse"]), rel_err(dImD[:NV, :NV], s["dimp_dq"]), rel_err(K[:NV, :NV], s["dimp_ddv"]),
                    float(np.abs(dImD[:NV, NV:]).max()))      # (no velocity in the impulse dynamics)
            worst = max(worst, *errs)
            assert max(errs) < 1e-11, errs
    print("ImD, dImD/dq, M on the impulse stage against the independent vectors: worst %.2e" % worst)
