#!/usr/bin/env python3
"""Golden vectors for the FIXED-BASE path: the Newton direction of an iiwa14 problem from a DENSE solve of the whole horizon's un-condensed KKT system
(gen_golden_kkt.py's assembly with the arm's dimensions: nv = nu = 7, no passive rows, no contacts, no events).

Until round 6 the condensation + Riccati + expansion of UnOCPSolver (unconstrained_dynamics.hxx:55-106, split_unriccati_factorizer.hxx, unriccati_recursion.cpp)
were held by the dense-formula identities of the reference's unit tests, restated in tests/test_oracle_unocp.py -- identities the restatement also
implements.  This fixture shares no formula with them: the stage data are captured by the oracle's OCPSolver restatement on the arm (oracle/ocp.cpp with
no passive rows) just before ITS condensation, assembled into one dense symmetric system in all of (dq dv da du | dlmd dgmm dbeta) and solved with
iterative refinement.  tests/test_golden_kkt.py holds THREE things to it at 1e-9: the oracle's OCPSolver on the arm (condensation through M^-1 + Riccati on u),
the oracle's UnOCPSolver (condensation through u = ID + Riccati on a) and, with -m gpu, the HIP kernels.

Output: tests/golden/kkt_iiwa14.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_kkt as G  # noqa: E402

FIELDS = ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta")


def dense_direction(o, M, dq0, dv0):
    """gen_golden_kkt.dense_direction with the arm's dimensions (set for the call only: the ANYmal fixture's tests share the module)"""
    keep = (G.NV, G.NU, G.NX, G.NC3)
    G.NV, G.NU, G.NX, G.NC3 = 7, 7, 14, 1
    try:
        return G.dense_direction(o, M, dq0, dv0)
    finally:
        G.NV, G.NU, G.NX, G.NC3 = keep


def problem_spec():
    rng = np.random.default_rng(7007)
    nv = 7
    lw = lambda lo, hi: (10.0 ** rng.uniform(lo, hi, nv)).tolist()
    return {"N": 6, "T": 0.3,
            "q_ref": rng.uniform(-1, 1, nv).tolist(), "v_ref": rng.uniform(-1, 1, nv).tolist(), "u_ref": rng.uniform(-3, 3, nv).tolist(),
            "q_weight": lw(0, 1.5), "qf_weight": lw(0, 1.5), "v_weight": lw(-1.5, 0), "vf_weight": lw(-1.5, 0), "a_weight": lw(-2.5, -1), "u_weight": lw(-4, -2.5),
            "u_max": rng.uniform(60, 200, nv).tolist(),
            "q": rng.uniform(-0.8, 0.8, nv).tolist(), "v": rng.uniform(-0.3, 0.3, nv).tolist(), "a": rng.uniform(-0.5, 0.5, nv).tolist(),
            "u": rng.uniform(-5, 5, nv).tolist(),
            "q_meas": rng.uniform(-0.8, 0.8, nv).tolist(), "v_meas": rng.uniform(-0.3, 0.3, nv).tolist()}


def problem(spec):
    import helpers as H
    from idocp_amd import capi
    m = H.iiwa14_model()
    cost = capi.Cost()
    for k in ("q_ref", "v_ref", "u_ref", "q_weight", "qf_weight", "v_weight", "vf_weight", "a_weight", "u_weight"):
        cost.set(k, np.array(spec[k]))
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    for i in range(m.nu):
        m.u_max[i] = spec["u_max"][i]
    return m, cost, cons


def build(spec, Solver, **kw):
    """(solver at the start iterate, measured q, measured v): the same calls for OracleOCP, OracleUnOCP and HipUnOCP"""
    m, cost, cons = problem(spec)
    o = Solver(m, cost, cons, spec["T"], spec["N"], **kw)
    for name in ("q", "v", "a", "u"):
        o.set_solution(name, np.array(spec[name]))
    if hasattr(o, "init_constraints"):
        o.init_constraints(0.0)
    return o, np.array(spec["q_meas"]), np.array(spec["v_meas"])


def main():
    import helpers as H
    spec = problem_spec()
    o, qm, vm = build(spec, H.OracleOCP)
    M = spec["N"] + 1
    assert o.update(0.0, qm, vm) == 0                      # one Newton iteration first: multipliers, slacks and duals off their start values
    o.lib.oracle_ocp_keep_uncondensed.argtypes = [C.c_void_p, C.c_int]
    o.lib.oracle_ocp_keep_uncondensed(o.h, 1)
    assert o.update(0.0, qm, vm) == 0
    orc = {f: o.get(f) for f in FIELDS}
    dense, nodes, info = dense_direction(o, M, orc["dq"][0], orc["dv"][0])
    print("dense system:", info)
    worst = 0.0
    for f in FIELDS:
        n = orc[f].shape[0]
        scale = max(1.0, np.max(np.abs(dense[f])))
        err = np.max(np.abs(dense[f][:n] - orc[f])) / scale
        worst = max(worst, err)
        print("  %-8s max |dense| %.3e   oracle OCPSolver - dense %.2e" % (f, np.max(np.abs(dense[f])), err))
    assert worst < 1e-8
    out = {"_about": "tests/golden/gen_golden_kkt_iiwa14.py: Newton direction of an iiwa14 problem from a dense solve of the un-condensed KKT system of the whole horizon",
           "spec": spec, "dense_system": info, "direction": {f: dense[f].tolist() for f in FIELDS}}
    with open(os.path.join(HERE, "kkt_iiwa14.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote kkt_iiwa14.json (%d nodes, %d unknowns)" % (M, info["unknowns"]))


if __name__ == "__main__":
    main()
