"""The checker of tests/test_other_quadrupeds_gpu.py, checked: the oracle's rigid-body terms on the mutated quadrupeds (random joint axes, rotated
placements and contact frames, other inertias) against the independent restatement of tests/golden/gen_golden_rbd.py evaluated LIVE on the same model --
body-frame recursive Newton-Euler with complex-step derivatives, no formula shared with the oracle's world-frame analytic derivatives.  CPU only."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, P, arr, oracle, rel_err
from test_other_quadrupeds_gpu import other_quadruped

sys.path.insert(0, GOLDEN)
import gen_golden_rbd as G      # noqa: E402


def generator_model(m):
    """The generator's model dictionary (gen_golden_rbd.load_model) from the C struct."""
    n = m.njoints
    M = dict(parent=[m.parent[i] for i in range(n)], jtype=[m.jtype[i] for i in range(n)], idx_q=[m.idx_q[i] for i in range(n)],
             idx_v=[m.idx_v[i] for i in range(n)], axis=[np.array(m.axis[i][:]) for i in range(n)],
             plc_R=[np.array(m.plc_R[i][:]).reshape(3, 3) for i in range(n)], plc_p=[np.array(m.plc_p[i][:]) for i in range(n)],
             body=[(m.mass[i], np.array(m.com[i][:]), np.array(m.inertia[i][:]).reshape(3, 3)) for i in range(n)],      # (inertia about the centre of mass, both)
             nq=m.nq, nv=m.nv, floating=m.has_floating_base, njoints=n, gravity=np.array(m.gravity[:]))
    M["contacts"] = [(m.contact_frame_id[c], m.contact_joint[c], np.array(m.contact_R[c][:]).reshape(3, 3), np.array(m.contact_p[c][:]))
                     for c in range(m.ncontacts)]
    return M


def test_the_conversion_reproduces_the_committed_anymal_vectors():
    """(generator_model is right: on the unmutated model the live generator gives the committed fixture)"""
    import json
    from helpers import anymal_model
    with open(os.path.join(GOLDEN, "rbd_anymal.json")) as f:
        g = json.load(f)
    M = generator_model(anymal_model())
    s = g["samples"][0]
    fext = np.zeros((M["njoints"], 6))
    for c, (_, jid, Rc, pc) in enumerate(M["contacts"]):
        fl = Rc @ np.array(s["f"][c])
        fext[jid, :3] += fl
        fext[jid, 3:] += np.cross(pc, fl)
    assert rel_err(G.rnea(M, arr(s["q"]), arr(s["v"]), arr(s["a"]), fext), s["tau"]) < 1e-13


@pytest.mark.parametrize("seed", [0, 3, 11, 20])
def test_oracle_rnea_and_derivatives_on_a_mutated_quadruped(seed):
    m, _ = other_quadruped(seed)
    M = generator_model(m)
    rng = np.random.default_rng(seed)
    ol, nv = oracle(), m.nv
    for _ in range(3):
        q, v, a = G.random_q(M, rng), rng.uniform(-1, 1, nv), rng.uniform(-1, 1, nv)
        fc = rng.uniform(-20, 40, (m.ncontacts, 3))
        fext = np.zeros((M["njoints"], 6))
        for c, (_, jid, Rc, pc) in enumerate(M["contacts"]):
            fl = Rc @ fc[c]
            fext[jid, :3] += fl
            fext[jid, 3:] += np.cross(pc, fl)
        f = arr(fc).reshape(-1)
        tau, dq, dv, da = np.zeros(nv), np.zeros((nv, nv)), np.zeros((nv, nv)), np.zeros((nv, nv))
        ol.oracle_rnea(C.byref(m), P(q), P(v), P(a), P(f), 1, P(tau))
        ol.oracle_rnea_derivatives(C.byref(m), P(q), P(v), P(a), P(f), 1, P(dq), P(dv), P(da))
        gq, gv, ga = G.rnea_derivatives(M, q, v, a, fext)
        assert rel_err(tau, G.rnea(M, q, v, a, fext)) < 1e-13
        assert rel_err(dq.T, gq) < 1e-12 and rel_err(dv.T, gv) < 1e-12 and rel_err(da.T, ga) < 1e-12
