#!/usr/bin/env python3
"""Golden vectors for the CONDENSATION + EXPANSION layer: the Newton direction of a small hybrid ANYmal problem from a DENSE solve of the
whole horizon's un-condensed KKT system (numpy only; run in the build container).

What the other fixtures leave open (review of round 5): `rbd_*.json` pins the rigid-body terms and `riccati_lqr.json` the Riccati sweep, but
ContactDynamics::condenseContactDynamics / ImpulseDynamicsForwardEuler::condenseImpulseDynamics (contact_dynamics.hxx:105-158,
impulse_dynamics_forward_euler.hxx:59-105), the condensed state equation (state_equation.hxx:40-63), the switching-constraint Schur step and the
expansion of the direction (contact_dynamics.hxx:161-190, riccati_recursion_solver.cpp:174-251) were held only by identities the oracle also
implements.  This generator shares NO formula with them.  It asks the oracle for the un-condensed stage data of every node of the chain -- cost +
barrier Hessians and gradients in all of (q, v, a, f, u), the raw state-equation Jacobians, [dID; dC] / d(q, v, a), [ID; C], the switching rows;
captured by OCPSolver::linearizeNode just BEFORE condensation (oracle/ocp.hpp UncondensedC) -- assembles the Newton system of the whole horizon
in all variables

    primal  dq dv (every node), da df du (every node but the terminal one)
    dual    dlmd dgmm (every node; node 0: the initial-state constraint), dbeta dmu (inverse dynamics, contact constraint), dxi (switching)

as one dense matrix (the reference's own Lagrangian: cost + lmd_{p+1}^T Fq_p + gmm_{p+1}^T Fv_p + dt beta^T (ID - S^T u) + dt mu^T C + xi^T P,
Gauss-Newton Hessian as in idocp), solves it with iterative refinement in extended precision, and writes the direction field by field.
`tests/test_golden_kkt.py` holds the oracle's condense -> Riccati -> expand direction to 1e-9 of it, the `-m gpu` twin the HIP direction.
(test/ocp/contact_dynamics_test.cpp and riccati_recursion_solver_test.cpp of the reference check their layers the same way, against dense formulas.)

The problem: ANYmal, N = 4, dt = 0.05, two feet on the ground, touch-down of the other two at t = 0.07 (an impulse + an aux stage, a switching
constraint in front of them), lift-off of two feet at t = 0.16 (a lift stage): 2-contact stages, 4-contact stages, an impulse stage, every
node kind of the hybrid chain.  The iterate is moved off the trivial start by one Newton iteration first, the measured state differs from the
iterate's, so every block of the system is populated.

Output: tests/golden/kkt_anymal.json -- the problem (as data), the chain, and the dense direction."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

NV, NU, NX = 18, 12, 36      # (gen_golden_kkt_iiwa14.py sets them to the arm's 7, 7, 14 and runs the same assembly)
NC3 = 12                      # width of the per-contact fields f / mu / xi
FIELDS = ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu", "dnu_passive", "dxi")


def problem_spec():
    """The problem as data (the tests rebuild it from this)."""
    rng = np.random.default_rng(20256)
    return {"N": 4, "T": 0.2, "max_num_impulse": 2,
            "status0": [0, 1, 1, 0], "events": [{"active": [1, 1, 1, 1], "t": 0.07}, {"active": [1, 0, 0, 1], "t": 0.16}],
            "q_joint_offset": (0.05 * rng.uniform(-1, 1, 12)).tolist(), "v": (0.2 * rng.uniform(-1, 1, NV)).tolist(),
            "a": (0.5 * rng.uniform(-1, 1, NV)).tolist(), "u": (2.0 * rng.uniform(-1, 1, NU)).tolist(),
            "q_meas_joint_offset": (0.02 * rng.uniform(-1, 1, 12)).tolist(), "q_meas_base_xy": (0.01 * rng.uniform(-1, 1, 2)).tolist(),
            "v_meas": (0.1 * rng.uniform(-1, 1, NV)).tolist()}


def build(spec, Solver, **kw):
    """(solver at the start iterate, measured q, measured v) -- the same calls for the oracle and the HIP wrapper"""
    import helpers as H
    m = H.anymal_model()
    cost, cons = H.anymal_problem(m, trotting_ref=True)
    o = Solver(m, cost, cons, spec["T"], spec["N"], max_num_impulse=spec["max_num_impulse"], **kw)
    pts = H.anymal_contact_points(m)
    o.set_contact_status(spec["status0"], pts)
    for ev in spec["events"]:
        o.push_back_contact_status(ev["active"], pts, ev["t"])
    q = H.ANYMAL_Q_STANDING.copy()
    q[7:] += np.array(spec["q_joint_offset"])
    o.set_solution("q", q)
    o.set_solution("v", np.array(spec["v"]))
    o.set_solution("a", np.array(spec["a"]))
    o.set_solution("u", np.array(spec["u"]))
    o.set_solution("f", [0, 0, 0.5 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    qm = q.copy()
    qm[7:] += np.array(spec["q_meas_joint_offset"])
    qm[0:2] += np.array(spec["q_meas_base_xy"])
    return o, qm, np.array(spec["v_meas"])


def fetch(o, pos, name, shape=None):
    lib = o.lib
    lib.oracle_ocp_get_uncondensed.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
    n = lib.oracle_ocp_get_uncondensed(o.h, pos, name.encode(), None)
    assert n >= 0, name
    out = np.zeros(max(n, 1))
    lib.oracle_ocp_get_uncondensed(o.h, pos, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)))
    out = out[:n]
    return out if shape is None else out.reshape(shape, order="F")


def stage_data(o, pos):
    meta = fetch(o, pos, "meta")
    assert meta[0] == 1.0, "no un-condensed record at chain position %d" % pos
    d = {"kind": int(meta[1]), "dimf": int(meta[2]), "dimi": int(meta[3]), "has_u": int(meta[4]), "dt": meta[5], "dtq": meta[6],
         "active": [(int(meta[7]) >> c) & 1 for c in range(4)]}
    nf, ni = d["dimf"], d["dimi"]
    d["Qxx"] = fetch(o, pos, "Qxx", (NX, NX))
    d["lq"], d["lv"] = fetch(o, pos, "lq"), fetch(o, pos, "lv")
    NP = NV - NU                  # passive rows of a floating base: the blocks of the configuration-difference Jacobians
    d["Fqq_prev"] = fetch(o, pos, "Fqq_prev", (NP, NP))
    if d["kind"] == 4:      # terminal
        return d
    d["Qaa"], d["Qff"] = fetch(o, pos, "Qaa"), fetch(o, pos, "Qff", (nf, nf))
    d["Quu"] = fetch(o, pos, "Quu", (NV, NV))
    for n in ("la", "lf", "lu", "lu_passive", "Fq", "Fv", "IDC"):
        d[n] = fetch(o, pos, n)
    d["Fqq"] = fetch(o, pos, "Fqq", (NP, NP))
    d["dIDCdqv"] = fetch(o, pos, "dIDCdqv", (NV + nf, NX))
    d["M"], d["J"] = fetch(o, pos, "M", (NV, NV)), fetch(o, pos, "J", (nf, NV))
    if ni:
        d["Phix"], d["Phia"], d["P"] = fetch(o, pos, "Phix", (ni, NX)), fetch(o, pos, "Phia", (ni, NV)), fetch(o, pos, "P")
    return d


def full(block6, tail):
    """nv x nv Jacobian of the floating base's configuration difference: the 6 x 6 base block, `tail` * I on the joints"""
    A = tail * np.eye(NV)
    NP = NV - NU
    A[:NP, :NP] = block6
    return A


def assemble(nodes, dq0, dv0):
    """The dense Newton system K z = r of the whole chain (module docstring).  Returns K, r and the index maps."""
    idx, n = {}, 0

    def alloc(key, size):
        nonlocal n
        idx[key] = slice(n, n + size)
        n += size

    M = len(nodes)
    for p, d in enumerate(nodes):
        alloc(("q", p), NV); alloc(("v", p), NV)
        if d["kind"] != 4:
            alloc(("a", p), NV); alloc(("f", p), d["dimf"])
            if d["has_u"]:
                alloc(("u", p), NU)
    for p, d in enumerate(nodes):
        alloc(("lmd", p), NV); alloc(("gmm", p), NV)
        if d["kind"] != 4:
            alloc(("beta", p), NV); alloc(("mu", p), d["dimf"])
            if d["dimi"]:
                alloc(("xi", p), d["dimi"])
    K, r = np.zeros((n, n)), np.zeros(n)
    # the ROW of a stationarity condition shares the index of its variable, the row of a constraint that of its multiplier
    for p, d in enumerate(nodes):
        q, v, lm, gm = idx[("q", p)], idx[("v", p)], idx[("lmd", p)], idx[("gmm", p)]
        Bq = full(d["Fqq_prev"], -1.0)                      # d Fq_{p-1} / d q_p = dSubtract_dMinus(q_{p-1}, q_p)
        K[q, q] += d["Qxx"][:NV, :NV]; K[q, v] += d["Qxx"][:NV, NV:]
        K[v, q] += d["Qxx"][NV:, :NV]; K[v, v] += d["Qxx"][NV:, NV:]
        K[q, lm] += Bq.T
        K[v, gm] += -np.eye(NV)
        r[q], r[v] = -d["lq"], -d["lv"]
        if p == 0:
            # the initial state is prescribed: dq_0 = q_meas (-) q_0, dv_0 = v_meas - v_0 (riccati_recursion_solver.cpp:110-118); written
            # with the Jacobians the multipliers enter rows q_0 / v_0 with, so that K stays symmetric
            K[lm, q] += Bq; r[lm] = Bq @ dq0
            K[gm, v] += -np.eye(NV); r[gm] = -dv0
        if d["kind"] == 4:
            continue
        dt, dtq, nf = d["dt"], d["dtq"], d["dimf"]
        a, f, be, mu = idx[("a", p)], idx[("f", p)], idx[("beta", p)], idx[("mu", p)]
        lm1, gm1, q1, v1 = idx[("lmd", p + 1)], idx[("gmm", p + 1)], idx[("q", p + 1)], idx[("v", p + 1)]
        Aq = full(d["Fqq"], 1.0)                            # d Fq_p / d q_p = dSubtract_dPlus(q_p, q_{p+1})
        Bq1 = full(nodes[p + 1]["Fqq_prev"], -1.0)
        dIDdq, dIDdv = d["dIDCdqv"][:NV, :NV], d["dIDCdqv"][:NV, NV:]
        dCdq, dCdv = d["dIDCdqv"][NV:, :NV], d["dIDCdqv"][NV:, NV:]
        # stationarity
        K[q, lm1] += Aq.T; K[q, be] += dt * dIDdq.T; K[q, mu] += dt * dCdq.T
        K[v, lm1] += dtq * np.eye(NV); K[v, gm1] += np.eye(NV); K[v, be] += dt * dIDdv.T; K[v, mu] += dt * dCdv.T
        K[a, a] += np.diag(d["Qaa"]); K[a, gm1] += dt * np.eye(NV); K[a, be] += dt * d["M"].T; K[a, mu] += dt * d["J"].T
        r[a] = -d["la"]
        K[f, f] += d["Qff"]; K[f, be] += -dt * d["J"]
        r[f] = -d["lf"]
        if d["has_u"]:
            u = idx[("u", p)]
            NP = NV - NU
            S = np.zeros((NU, NV)); S[:, NP:] = np.eye(NU)                 # u enters the actuated rows of ID
            K[u, u] += d["Quu"][NP:, NP:]; K[u, be] += -dt * S
            r[u] = -d["lu"]
            K[be, u] += -dt * S.T
        # state equation p -> p + 1 (multipliers lmd_{p+1}, gmm_{p+1})
        K[lm1, q] += Aq; K[lm1, v] += dtq * np.eye(NV); K[lm1, q1] += Bq1
        r[lm1] = -d["Fq"]
        K[gm1, v] += np.eye(NV); K[gm1, a] += dt * np.eye(NV); K[gm1, v1] += -np.eye(NV)
        r[gm1] = -d["Fv"]
        # inverse dynamics and contact constraint (scaled by dt like their multipliers' columns)
        K[be, q] += dt * dIDdq; K[be, v] += dt * dIDdv; K[be, a] += dt * d["M"]; K[be, f] += -dt * d["J"].T
        r[be] = -dt * d["IDC"][:NV]
        K[mu, q] += dt * dCdq; K[mu, v] += dt * dCdv; K[mu, a] += dt * d["J"]
        r[mu] = -dt * d["IDC"][NV:]
        if d["dimi"]:
            xi = idx[("xi", p)]
            Phiq, Phiv = d["Phix"][:, :NV], d["Phix"][:, NV:]
            K[q, xi] += Phiq.T; K[v, xi] += Phiv.T; K[a, xi] += d["Phia"].T
            K[xi, q] += Phiq; K[xi, v] += Phiv; K[xi, a] += d["Phia"]
            r[xi] = -d["P"]
    return K, r, idx


def solve_refined(K, r, rounds=4):
    """LU solve + iterative refinement with the residual in extended precision (the barrier terms make K's condition number ~1e10)"""
    import scipy.linalg as sl
    lu = sl.lu_factor(K)
    z = sl.lu_solve(lu, r)
    Kl, rl = K.astype(np.longdouble), r.astype(np.longdouble)
    for _ in range(rounds):
        res = (rl - Kl @ z.astype(np.longdouble)).astype(np.float64)
        z = z + sl.lu_solve(lu, res)
    res = (rl - Kl @ z.astype(np.longdouble)).astype(np.float64)
    return z, float(np.max(np.abs(res)))


def dense_direction(o, M, dq0, dv0):
    nodes = [stage_data(o, p) for p in range(M)]
    K, r, idx = assemble(nodes, dq0, dv0)
    assert np.max(np.abs(K - K.T)) < 1e-9 * np.max(np.abs(K)), "the Newton system of an equality-constrained problem is symmetric"
    z, res = solve_refined(K, r)
    NP = NV - NU
    out = {f: np.zeros((M, dim)) for f, dim in (("dq", NV), ("dv", NV), ("da", NV), ("df", NC3), ("du", NU), ("dlmd", NV), ("dgmm", NV),
                                                  ("dbeta", NV), ("dmu", NC3), ("dnu_passive", NP), ("dxi", NC3))}
    newton = {"dbeta": np.zeros((M, NV))}      # the exact Newton dbeta, for the record (differs from the reference's on switching stages)
    for p, d in enumerate(nodes):
        out["dq"][p], out["dv"][p] = z[idx[("q", p)]], z[idx[("v", p)]]
        out["dlmd"][p], out["dgmm"][p] = z[idx[("lmd", p)]], z[idx[("gmm", p)]]
    for p, d in enumerate(nodes):
        if d["kind"] == 4:
            continue
        out["da"][p] = z[idx[("a", p)]]
        dbeta, dmu = z[idx[("beta", p)]], z[idx[("mu", p)]]
        newton["dbeta"][p] = dbeta
        if d["dimi"]:
            # THE REFERENCE'S dual direction on a stage that carries a switching constraint is not the Newton step: ContactDynamics::
            # computeCondensedDualDirection (contact_dynamics.hxx:171-190) forms [dbeta; dmu] from la + dt dgmm_next and lf, i.e. it solves rows a
            # and f of the stationarity conditions WITHOUT the term Phia^T dxi that the switching constraint adds to row a (split_ocp.hxx:124-131
            # puts Phia^T xi into la, nothing carries dxi).  Everything else -- dq dv da df du dlmd dgmm dxi of every node -- IS the Newton
            # step (found by this fixture, round 6).  The golden vector follows the reference: rows a, f without that term, solved densely.
            nf = d["dimf"]
            A = np.block([[d["dt"] * d["M"].T, d["dt"] * d["J"].T], [-d["dt"] * d["J"], np.zeros((nf, nf))]])
            rhs = np.concatenate([-d["la"] - d["Qaa"] * out["da"][p] - d["dt"] * out["dgmm"][p + 1], -d["lf"] - d["Qff"] @ z[idx[("f", p)]]])
            sol = np.linalg.solve(A, rhs)
            sol = sol + np.linalg.solve(A, rhs - A @ sol)
            dbeta, dmu = sol[:NV], sol[NV:]
        out["dbeta"][p] = dbeta
        rows = [3 * c + k for c in range(4) if d["active"][c] for k in range(3)]      # contact c of the solver's 12-wide f / mu fields
        out["df"][p, rows], out["dmu"][p, rows] = z[idx[("f", p)]], dmu
        if d["has_u"]:
            out["du"][p] = z[idx[("u", p)]]
            # passive torques are fixed at zero; nu_passive is the multiplier of that: row u_passive of the stationarity conditions reads
            # - dt dbeta[:6] + dt dnu = - lu_passive
            out["dnu_passive"][p] = out["dbeta"][p, :NP] - d["lu_passive"] / d["dt"]
        if d["dimi"]:
            out["dxi"][p, :d["dimi"]] = z[idx[("xi", p)]]
    return out, nodes, {"unknowns": int(K.shape[0]), "max_abs_residual": res, "cond_estimate": float(np.linalg.cond(K)),
                        "newton_minus_reference_dbeta_on_switching_stages": float(np.max(np.abs(newton["dbeta"] - out["dbeta"])))}


def packed(o, name, M, chain):
    """the oracle's (or the HIP wrapper's) direction field along the chain; df / dmu / dxi packed over the ACTIVE contacts like the dense ones"""
    return o.get_chain(name, M)


def main():
    import helpers as H
    spec = problem_spec()
    o, qm, vm = build(spec, H.OracleOCP)
    chain = o.chain(0.0)
    M = len(chain)
    assert o.update(0.0, qm, vm) == 0                      # one Newton iteration: every multiplier, slack and dual is off its start value
    o.lib.oracle_ocp_keep_uncondensed.argtypes = [C.c_void_p, C.c_int]
    o.lib.oracle_ocp_keep_uncondensed(o.h, 1)
    assert o.update(0.0, qm, vm) == 0                      # the iteration under test (its direction stays readable behind the update)
    orc = {f: o.get_chain(f, M) for f in FIELDS}
    dense, nodes, info = dense_direction(o, M, orc["dq"][0], orc["dv"][0])
    print("chain:", [(c["kind"], c["dimf"], round(c["dt"], 4), c["sw_event"]) for c in chain])
    print("dense system:", info)
    worst = 0.0
    for f in FIELDS:
        scale = max(1.0, np.max(np.abs(dense[f])))
        err = np.max(np.abs(dense[f] - orc[f])) / scale
        worst = max(worst, err)
        print("  %-12s max |dense| %.3e   oracle - dense (relative to the field's largest entry) %.2e" % (f, np.max(np.abs(dense[f])), err))
    assert worst < 1e-8, "the oracle's direction is not the Newton step of the un-condensed system"
    out = {"_about": "tests/golden/gen_golden_kkt.py: Newton direction of a hybrid ANYmal problem from a dense solve of the un-condensed KKT system of the whole horizon",
           "spec": spec, "chain": [{"kind": c["kind"], "dimf": int(c["dimf"]), "dt": float(c["dt"]), "sw_event": int(c["sw_event"])} for c in chain],
           "dense_system": info, "direction": {f: dense[f].tolist() for f in FIELDS}}
    with open(os.path.join(HERE, "kkt_anymal.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote kkt_anymal.json (%d nodes, %d unknowns)" % (M, info["unknowns"]))


if __name__ == "__main__":
    main()
