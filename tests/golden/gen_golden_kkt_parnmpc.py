#!/usr/bin/env python3
"""Golden vectors for the ParNMPC stage: the stage-wise Newton step of a backward-Euler ANYmal stage from a DENSE solve of its un-condensed KKT system.

ParNMPCSolver's coarse update (split_backward_correction.hxx:50-82) is the Newton step of ONE stage's subproblem -- the stage's cost, its backward-Euler state
equation against the stage before it, the inverse dynamics and the contact constraint, the next stage's influence folded into `aux_mat` -- computed by condensing
the contact dynamics (contact_dynamics.hxx:105-158 with the backward-Euler signs, split_parnmpc.hxx) and inverting the condensed KKT matrix block-wise
(split_kkt_matrix_inverter.hxx:56-197).  On a horizon of ONE stage there is nothing else: no neighbour, no correction sweep, the direction of updateSolution IS that
step.  This generator takes the un-condensed data of that stage from the oracle (ParNMPCSolver::keep_uncondensed, captured before condensation), assembles the full
system in (dq dv da df du | dlmd dgmm dbeta dmu) and solves it densely; tests/test_golden_kkt.py holds the oracle's condensed route and, with -m gpu, the HIP kernels
(K5<BWD>, K9w, K6 / K7) to it at 1e-9.  Three contact configurations: all four feet, two feet, none.

Output: tests/golden/kkt_parnmpc.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import gen_golden_kkt as G  # noqa: E402

NV, NU, NX = 18, 12, 36
FIELDS = ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu", "dnu_passive")
CASES = {"four_feet": [1, 1, 1, 1], "two_feet": [0, 1, 1, 0], "flight": [0, 0, 0, 0]}


def problem_spec():
    rng = np.random.default_rng(8128)
    return {"T": 0.04, "q_joint_offset": (0.05 * rng.uniform(-1, 1, 12)).tolist(), "v": (0.2 * rng.uniform(-1, 1, NV)).tolist(),
            "a": (0.5 * rng.uniform(-1, 1, NV)).tolist(), "u": (2.0 * rng.uniform(-1, 1, NU)).tolist(),
            "q_meas_joint_offset": (0.02 * rng.uniform(-1, 1, 12)).tolist(), "q_meas_base": (0.01 * rng.uniform(-1, 1, 3)).tolist(),
            "v_meas": (0.1 * rng.uniform(-1, 1, NV)).tolist()}


def build(spec, active, Solver, **kw):
    import helpers as H
    m = H.anymal_model()
    cost, cons = H.anymal_problem(m, trotting_ref=False)
    o = Solver(m, cost, cons, spec["T"], 1, **kw)
    o.set_contact_status(active, H.anymal_contact_points(m))
    q = H.ANYMAL_Q_STANDING.copy()
    q[7:] += np.array(spec["q_joint_offset"])
    o.set_solution("q", q)
    o.set_solution("v", np.array(spec["v"]))
    o.set_solution("a", np.array(spec["a"]))
    o.set_solution("u", np.array(spec["u"]))
    o.set_solution("f", [0, 0, 0.3 * (-m.total_mass * m.gravity[2])])
    o.init(0.0)
    qm = q.copy()
    qm[7:] += np.array(spec["q_meas_joint_offset"])
    qm[0:3] += np.array(spec["q_meas_base"])
    return o, qm, np.array(spec["v_meas"])


def fetch(o, name, shape=None):
    lib = o.lib
    lib.oracle_parnmpc_get_uncondensed.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
    n = lib.oracle_parnmpc_get_uncondensed(o.h, 0, name.encode(), None)
    assert n >= 0, name
    out = np.zeros(max(n, 1))
    lib.oracle_parnmpc_get_uncondensed(o.h, 0, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)))
    out = out[:n]
    return out if shape is None else out.reshape(shape, order="F")


def dense_stage_step(o):
    """The Newton step of the one-stage problem from its un-condensed data.  Rows: the stationarity condition of a variable shares its index, a constraint
    that of its multiplier (symmetric system)."""
    meta = fetch(o, "meta")
    assert meta[0] == 1.0
    nf, dt = int(meta[2]), meta[5]
    active = [(int(meta[7]) >> c) & 1 for c in range(4)]
    Qxx = fetch(o, "Qxx", (NX, NX))
    Qaa, Qff, Quu = fetch(o, "Qaa"), fetch(o, "Qff", (nf, nf)), fetch(o, "Quu", (NV, NV))
    lq, lv, la, lf, lu, lup = (fetch(o, n) for n in ("lq", "lv", "la", "lf", "lu", "lu_passive"))
    Fq, Fv, IDC = fetch(o, "Fq"), fetch(o, "Fv"), fetch(o, "IDC")
    Fqq = -np.eye(NV)
    Fqq[:6, :6] = fetch(o, "Fqq", (6, 6))                  # d Fq / d q = dSubtract_dMinus(q_prev, q): the base block, -I on the joints
    dIDC = fetch(o, "dIDCdqv", (NV + nf, NX))
    Mm, J = fetch(o, "M", (NV, NV)), fetch(o, "J", (nf, NV))
    dIDdq, dIDdv, dCdq, dCdv = dIDC[:NV, :NV], dIDC[:NV, NV:], dIDC[NV:, :NV], dIDC[NV:, NV:]
    idx, n = {}, 0
    for key, size in (("q", NV), ("v", NV), ("a", NV), ("f", nf), ("u", NU), ("lmd", NV), ("gmm", NV), ("beta", NV), ("mu", nf)):
        idx[key] = slice(n, n + size)
        n += size
    K, r = np.zeros((n, n)), np.zeros(n)
    q, v, a, f, u, lm, gm, be, mu = (idx[k] for k in ("q", "v", "a", "f", "u", "lmd", "gmm", "beta", "mu"))
    I = np.eye(NV)
    S = np.zeros((NU, NV)); S[:, 6:] = np.eye(NU)
    K[q, q] += Qxx[:NV, :NV]; K[q, v] += Qxx[:NV, NV:]; K[v, q] += Qxx[NV:, :NV]; K[v, v] += Qxx[NV:, NV:]
    K[a, a] += np.diag(Qaa); K[f, f] += Qff; K[u, u] += Quu[6:, 6:]
    # backward-Euler state equation of the stage: Fq = (q_prev (-) q) + dt v, Fv = v_prev - v + dt a (state_equation.hxx:111-147)
    K[lm, q] += Fqq; K[lm, v] += dt * I; K[gm, v] += -I; K[gm, a] += dt * I
    # inverse dynamics and contact constraint, scaled by dt like their multipliers' columns
    K[be, q] += dt * dIDdq; K[be, v] += dt * dIDdv; K[be, a] += dt * Mm; K[be, f] += -dt * J.T; K[be, u] += -dt * S.T
    K[mu, q] += dt * dCdq; K[mu, v] += dt * dCdv; K[mu, a] += dt * J
    for row, col in ((lm, q), (lm, v), (gm, v), (gm, a), (be, q), (be, v), (be, a), (be, f), (be, u), (mu, q), (mu, v), (mu, a)):
        K[col, row] += K[row, col].T
    r[q], r[v], r[a], r[f], r[u] = -lq, -lv, -la, -lf, -lu
    r[lm], r[gm], r[be], r[mu] = -Fq, -Fv, -dt * IDC[:NV], -dt * IDC[NV:]
    assert np.max(np.abs(K - K.T)) < 1e-12 * np.max(np.abs(K))
    z, res = G.solve_refined(K, r)
    out = {"dq": z[q], "dv": z[v], "da": z[a], "du": z[u], "dlmd": z[lm], "dgmm": z[gm], "dbeta": z[be],
           "df": np.zeros(12), "dmu": np.zeros(12), "dnu_passive": z[be][:6] - lup / dt}
    rows = [3 * c + k for c in range(4) if active[c] for k in range(3)]
    out["df"][rows], out["dmu"][rows] = z[f], z[mu]
    return out, {"unknowns": int(n), "max_abs_residual": res, "cond_estimate": float(np.linalg.cond(K)), "dimf": nf}


def run(spec, active, Solver, **kw):
    o, qm, vm = build(spec, active, Solver, **kw)
    assert o.update(0.0, qm, vm) == 0                      # one iteration first: multipliers, slacks and duals off their start values
    return o, qm, vm


def main():
    import helpers as H
    spec = problem_spec()
    out = {"_about": "tests/golden/gen_golden_kkt_parnmpc.py: stage-wise Newton step of a backward-Euler ANYmal stage (ParNMPC, N = 1) from a dense solve of its un-condensed KKT system",
           "spec": spec, "cases": {}}
    for name, active in CASES.items():
        o, qm, vm = run(spec, active, H.OracleParNMPC)
        o.lib.oracle_parnmpc_keep_uncondensed.argtypes = [C.c_void_p, C.c_int]
        o.lib.oracle_parnmpc_keep_uncondensed(o.h, 1)
        assert o.update(0.0, qm, vm) == 0
        dense, info = dense_stage_step(o)
        worst = 0.0
        print(name, info)
        for f in FIELDS:
            have = o.get(f)[0]
            err = np.max(np.abs(have - dense[f])) / max(1.0, np.max(np.abs(dense[f])))
            worst = max(worst, err)
            print("  %-12s max |dense| %.3e   oracle - dense %.2e" % (f, np.max(np.abs(dense[f])), err))
        assert worst < 1e-8, (name, worst)
        out["cases"][name] = {"active": active, "dense_system": info, "direction": {f: dense[f].tolist() for f in FIELDS}}
    with open(os.path.join(HERE, "kkt_parnmpc.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote kkt_parnmpc.json")


if __name__ == "__main__":
    main()
