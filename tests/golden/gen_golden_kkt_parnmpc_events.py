#!/usr/bin/env python3
"""Golden vectors for the ParNMPC stages of a horizon WITH discrete events: the coarse update of every stage of the chain -- regular, aux (the stage in
front of an impulse, which carries the switching constraint), impulse, lift, terminal -- from a DENSE solve of that stage's un-condensed KKT system.

BackwardCorrectionSolver::coarseUpdate (backward_correction_solver.cpp:95-250) gives every stage of the chain the Newton step of ITS subproblem: the stage's
cost, its backward-Euler state equation against the stage before it (whose state is held fixed), its dynamics and constraints, and the influence of the stage
behind it folded into the Hessian as aux_mat.  The reference condenses the dynamics (contact_dynamics.hxx:105-158 with the backward-Euler signs;
impulse_dynamics_backward_euler.hxx:59-97 on an impulse stage), premultiplies the base rows of the state equation (state_equation.hxx condenseBackwardEuler,
impulse_state_equation.hxx:86-111) and inverts the condensed KKT matrix block-wise (split_kkt_matrix_inverter.hxx:44-166 with and without the switching rows,
impulse_split_kkt_matrix_inverter.hxx:34-80).  This generator does none of that: it takes the UN-condensed data of a stage from the oracle
(ParNMPCSolver::keep_uncondensed), assembles the full system in

    regular / aux / lift / terminal:  (dq dv da df du | dlmd dgmm dbeta dmu [dxi])          impulse:  (dq dv ddv df | dlmd dgmm dbeta dmu)

and solves it densely.  What it writes is the coarse iterate s_new = s (+) step of every chain position (the fields the backward correction carries: lmd gmm
u q v, xi on an aux stage, f mu on an impulse stage).  Two pieces of Lie-group arithmetic connect the step to s_new and both come from the independent
restatement of tests/golden/gen_golden_rbd.py: q_new = integrate(q, dq), and the multiplier of the PREMULTIPLIED base rows, lmd_new[:6] = lmd[:6] + Jplus^T
dlmd[:6] with Jplus = d (q_prev (-) q) / d q_prev (extrapolated central differences of its own difference()).

tests/test_golden_kkt.py holds the oracle's condensed route and, with -m gpu, the HIP kernels (K5<BWD> and the event kernels, K9w in its regular and event
instantiations) to it at 1e-9.

Output: tests/golden/kkt_parnmpc_events.json"""
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
import gen_golden_rbd as RBD  # noqa: E402

NV, NU, NX, NQ = 18, 12, 36, 19      # (module globals: gen_golden_kkt_parnmpc_iiwa14.py runs the assembly below with the arm's)
FIELDS = ("new_lmd", "new_gmm", "new_q", "new_v", "new_u", "new_xi", "new_f", "new_mu")
# name: (contact status at the start, [(status after the event, event time in grid intervals)])
CASES = {
    "two_feet_land_then_two_lift": ([1, 0, 0, 1], [([1, 1, 1, 1], 2.4), ([0, 1, 1, 0], 5.6)]),      # 6-row impulse, then a lift
    "one_foot_lands": ([1, 1, 0, 1], [([1, 1, 1, 1], 3.3)]),                                        # 3 rows
    "landing_from_flight": ([0, 0, 0, 0], [([1, 1, 1, 1], 1.7)]),                                   # 12 rows
}
N, DT = 7, 0.035
# the gaits on which the whole-iteration comparison is made: ParNMPCSolver without a line search has to CONVERGE from this file's cold start for it (see
# whole_iteration); with two feet or none on the ground at the start it does not (the reference's examples warm-start such horizons)
WHOLE_CASES = {"one_foot_lands": CASES["one_foot_lands"], "standing": ([1, 1, 1, 1], [])}
WHOLE_AFTER = {"one_foot_lands": 10, "standing": 6}      # the comparison is made on the iteration behind this many: base step ~ 1e-6, above the rounding floor


def problem_spec():
    rng = np.random.default_rng(6174)
    return {"N": N, "T": N * DT, "q_joint_offset": (0.05 * rng.uniform(-1, 1, 12)).tolist(), "v": (0.2 * rng.uniform(-1, 1, NV)).tolist(),
            "a": (0.5 * rng.uniform(-1, 1, NV)).tolist(), "u": (2.0 * rng.uniform(-1, 1, NU)).tolist(),
            "q_meas_joint_offset": (0.02 * rng.uniform(-1, 1, 12)).tolist(), "q_meas_base": (0.01 * rng.uniform(-1, 1, 3)).tolist(),
            "v_meas": (0.1 * rng.uniform(-1, 1, NV)).tolist()}


def build(spec, case, Solver, **kw):
    import helpers as H
    m = H.anymal_model()
    cost, cons = H.anymal_problem(m, trotting_ref=False)
    first, events = CASES[case] if case in CASES else WHOLE_CASES[case]
    o = Solver(m, cost, cons, spec["T"], spec["N"], max_num_impulse=len(events), **kw)
    pts = H.anymal_contact_points(m)
    o.set_contact_status(first, pts)
    for status, when in events:
        o.push_back_contact_status(status, pts, when * spec["T"] / spec["N"])
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


def fetch(o, pos, name, shape=None):
    lib = o.lib
    lib.oracle_parnmpc_get_uncondensed.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
    n = lib.oracle_parnmpc_get_uncondensed(o.h, pos, name.encode(), None)
    assert n >= 0, name
    out = np.zeros(max(n, 1))
    lib.oracle_parnmpc_get_uncondensed(o.h, pos, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)))
    out = out[:n]
    return out if shape is None else out.reshape(shape, order="F")


def assemble_stage(o, pos):
    """The un-condensed KKT system of the subproblem of chain position `pos` (aux_mat of the stage behind in its Hessian, the stage in front held fixed).
    Rows: the stationarity condition of a variable shares its index, a constraint that of its multiplier (symmetric system)."""
    meta = fetch(o, pos, "meta")
    assert meta[0] == 1.0
    kind, nf, ni, has_u, dt = int(meta[1]), int(meta[2]), int(meta[3]), int(meta[4]), meta[5]
    impulse = has_u == 0
    active = [(int(meta[7]) >> c) & 1 for c in range(4)]
    Qxx = fetch(o, pos, "Qxx", (NX, NX)) + fetch(o, pos, "aux_next", (NX, NX))
    Qxx[NV:, :NV] = Qxx[:NV, NV:].T                        # (the carriers keep the upper blocks)
    Qaa, Qff = fetch(o, pos, "Qaa"), fetch(o, pos, "Qff", (nf, nf))
    lq, lv, la, lf = (fetch(o, pos, n) for n in ("lq", "lv", "la", "lf"))
    Fq, Fv, IDC = fetch(o, pos, "Fq"), fetch(o, pos, "Fv"), fetch(o, pos, "IDC")
    NP = NV - NU                                           # passive (floating-base) rows: 6 or none
    Fqq = -np.eye(NV)
    if NP:
        Fqq[:NP, :NP] = fetch(o, pos, "Fqq", (NP, NP))     # d (q_prev (-) q) / d q: the base block, -I on the joints
    dIDC = fetch(o, pos, "dIDCdqv", (NV + nf, NX))
    Mm, J = fetch(o, pos, "M", (NV, NV)), fetch(o, pos, "J", (nf, NV))
    dIDdq, dIDdv, dCdq, dCdv = dIDC[:NV, :NV], dIDC[:NV, NV:], dIDC[NV:, :NV], dIDC[NV:, NV:]
    nxi = ni if not impulse else 0
    idx, n = {}, 0
    for key, size in (("q", NV), ("v", NV), ("a", NV), ("f", nf), ("u", 0 if impulse else NU), ("lmd", NV), ("gmm", NV), ("beta", NV), ("mu", nf), ("xi", nxi)):
        idx[key] = slice(n, n + size)
        n += size
    K, r = np.zeros((n, n)), np.zeros(n)
    q, v, a, f, u, lm, gm, be, mu, xi = (idx[k] for k in ("q", "v", "a", "f", "u", "lmd", "gmm", "beta", "mu", "xi"))
    I = np.eye(NV)
    K[q, q] += Qxx[:NV, :NV]; K[q, v] += Qxx[:NV, NV:]; K[v, q] += Qxx[NV:, :NV]; K[v, v] += Qxx[NV:, NV:]
    K[a, a] += np.diag(Qaa); K[f, f] += Qff
    pairs = [(lm, q), (gm, v), (gm, a), (be, q), (be, a), (be, f), (mu, q), (mu, v)]
    if not impulse:
        Quu, lu = fetch(o, pos, "Quu", (NV, NV)), fetch(o, pos, "lu")
        S = np.zeros((NU, NV)); S[:, NP:] = np.eye(NU)
        K[u, u] += Quu[NP:, NP:]
        # backward-Euler state equation: Fq = (q_prev (-) q) + dt v, Fv = v_prev - v + dt a (state_equation.hxx:111-147)
        K[lm, q] += Fqq; K[lm, v] += dt * I; K[gm, v] += -I; K[gm, a] += dt * I
        # inverse dynamics and contact constraint, scaled by dt like their multipliers' columns
        K[be, q] += dt * dIDdq; K[be, v] += dt * dIDdv; K[be, a] += dt * Mm; K[be, f] += -dt * J.T; K[be, u] += -dt * S.T
        K[mu, q] += dt * dCdq; K[mu, v] += dt * dCdv; K[mu, a] += dt * J
        pairs += [(lm, v), (be, v), (be, u), (mu, a)]
        r[u] = -lu
        r[be], r[mu] = -dt * IDC[:NV], -dt * IDC[NV:]
        if nxi:                                             # switching constraint of the aux stage, on the configuration only (switching_constraint.hxx)
            K[xi, q] += fetch(o, pos, "Phix", (nxi, NV))
            pairs.append((xi, q))
            r[xi] = -fetch(o, pos, "P")
    else:
        # impulse stage: Fq = q_prev (-) q, Fv = v_prev - v + dv (impulse_state_equation.hxx:59-85); impulse dynamics M dv - J^T f = 0, the velocity of
        # the landing feet J v = 0 behind the impulse (impulse_dynamics_backward_euler.hxx:20-58): no time step anywhere
        K[lm, q] += Fqq; K[gm, v] += -I; K[gm, a] += I
        K[be, q] += dIDdq; K[be, a] += Mm; K[be, f] += -J.T
        K[mu, q] += dCdq; K[mu, v] += dCdv
        r[be], r[mu] = -IDC[:NV], -IDC[NV:]
    for row, col in pairs:
        K[col, row] += K[row, col].T
    r[q], r[v], r[a], r[f] = -lq, -lv, -la, -lf
    r[lm], r[gm] = -Fq, -Fv
    assert np.max(np.abs(K - K.T)) < 1e-12 * np.max(np.abs(K))
    rows = [3 * c + k for c in range(4) if active[c] for k in range(3)]
    return K, r, idx, rows, {"kind": kind, "unknowns": int(n), "dimf": nf, "switching_rows": nxi, "impulse": bool(impulse)}


def dense_stage_step(o, pos):
    """The Newton step of that subproblem."""
    K, r, idx, rows, info = assemble_stage(o, pos)
    z, res = G.solve_refined(K, r)
    info["max_abs_residual"] = res
    return {k: z[idx[k]] for k in idx}, rows, info


def jplus(model_dict, q, q_prev):
    """d (q_prev (-) q) / d q_prev, base block: central differences of the generator's own difference() at three step sizes, extrapolated to O(h^6)
    (gen_golden_rbd.ddifference stops at O(h^4): 1e-10, visible at this file's bar)."""
    if NV == NU:
        return np.zeros((0, 0))                            # no floating base: the configuration is a vector

    def central(h):
        J = np.zeros((6, 6))
        for k in range(6):
            e = np.zeros(NV)
            e[k] = h
            J[:, k] = (RBD.difference(model_dict, q, RBD.integrate(model_dict, q_prev, e)) - RBD.difference(model_dict, q, RBD.integrate(model_dict, q_prev, -e)))[:6] / (2 * h)
        return J
    d0, d1, d2 = central(8e-3), central(4e-3), central(2e-3)
    r0, r1 = (4 * d1 - d0) / 3, (4 * d2 - d1) / 3
    return (16 * r1 - r0) / 15


def coarse_iterates(o, model_dict, q_meas):
    """s_new of every chain position from the dense steps (see the header for the two Lie-group pieces)."""
    chain = o.chain(0.0)
    M = len(chain)
    cur = {f: o.get_chain(f, M) for f in ("lmd", "gmm", "q", "v", "u", "xi", "f", "mu")}
    out = {f: np.zeros_like(cur[f[4:]]) for f in FIELDS}
    infos = []
    for p in range(M):
        step, rows, info = dense_stage_step(o, p)
        q_prev = q_meas if p == 0 else cur["q"][p - 1]
        Jplus = jplus(model_dict, cur["q"][p], q_prev)
        dl = step["lmd"].copy()
        dl[:6] = Jplus.T @ dl[:6]
        out["new_lmd"][p] = cur["lmd"][p] + dl
        out["new_gmm"][p] = cur["gmm"][p] + step["gmm"]
        out["new_q"][p] = RBD.integrate(model_dict, cur["q"][p], step["q"])
        out["new_v"][p] = cur["v"][p] + step["v"]
        out["new_f"][p], out["new_mu"][p] = cur["f"][p], cur["mu"][p]      # (not touched by the coarse update of a stage that is not an impulse)
        if info["impulse"]:
            out["new_f"][p][rows] += step["f"]
            out["new_mu"][p][rows] += step["mu"]
        else:
            out["new_u"][p] = cur["u"][p] + step["u"]
        if info["switching_rows"]:
            out["new_xi"][p][:info["switching_rows"]] = cur["xi"][p][:info["switching_rows"]] + step["xi"]
        info["chain_kind"] = chain[p]["kind"]
        infos.append(info)
    return out, infos


DIRECTION = ("dq", "dv", "da", "df", "du", "dlmd", "dgmm", "dbeta", "dmu", "dxi")


def dense_iteration(o, model_dict, q_meas):
    """The direction of the WHOLE iteration (coarse update + the four correction sweeps + expansion) from one dense solve over the horizon.

    The backward correction is block back-substitution through the chain with the pivot of stage i taken as  K_i + aux_old_{i+1}  instead of the exact Schur
    complement (aux_old: BackwardCorrectionSolver::aux_mat_ as the LAST iteration left it): coarse update = the pivot's solve of the stage's own right-hand side,
    backward sweep = the change of the costate of the stage behind carried into it, forward sweep = the change of the state of the stage in front
    (backward_correction_solver.cpp:253-366).  A block back-substitution with given pivots solves SOME block-tridiagonal system exactly: the one whose
    diagonal blocks are  pivot_i + U_i pivot_{i+1}^-1 L_{i+1}  -- here the exact Newton matrix of the whole horizon plus, in the state block of every stage,
    aux_old_{i+1} - aux_new_{i+1}, where  aux_new_{i+1} = -L^T [pivot_{i+1}^-1]_(costate, costate) L  is what THIS iteration computes for the next one and
    L = d (state equation of stage i+1) / d x_i = diag(Jplus, I).  (At a fixed point of aux the two cancel and the iteration is Newton's.)  This function builds
    that matrix from the un-condensed stage blocks, the couplings between neighbours and dense inverses of the pivots, and solves it once."""
    M = len(o.chain(0.0)) if NV != NU else o.N
    qs = o.get_chain("q", M) if NV != NU else o.get("q")
    stages = [assemble_stage(o, p) for p in range(M)]
    off = np.concatenate([[0], np.cumsum([st[0].shape[0] for st in stages])])
    n = int(off[-1])
    A, b = np.zeros((n, n)), np.zeros(n)
    Ls = []
    for p, (K, r, idx, rows, info) in enumerate(stages):
        A[off[p]:off[p + 1], off[p]:off[p + 1]] = K
        b[off[p]:off[p + 1]] = r
        L = np.eye(NX)
        if NV != NU:
            L[:6, :6] = jplus(model_dict, qs[p], q_meas if p == 0 else qs[p - 1])
        Ls.append(L)
        if p > 0:                                          # rows (lmd, gmm) of this stage x columns (q, v) of the stage in front, and the transpose
            pidx = stages[p - 1][2]
            rl = np.r_[off[p] + idx["lmd"].start:off[p] + idx["lmd"].stop, off[p] + idx["gmm"].start:off[p] + idx["gmm"].stop]
            cx = np.r_[off[p - 1] + pidx["q"].start:off[p - 1] + pidx["q"].stop, off[p - 1] + pidx["v"].start:off[p - 1] + pidx["v"].stop]
            A[np.ix_(rl, cx)] += L
            A[np.ix_(cx, rl)] += L.T
    for p in range(M - 1):                                 # aux_old is already in K_p; take aux_new of the stage behind out
        Kn, _, nidx, _, _ = stages[p + 1]
        cc = np.r_[nidx["lmd"].start:nidx["lmd"].stop, nidx["gmm"].start:nidx["gmm"].stop]
        inv_cc = np.linalg.solve(Kn, np.eye(Kn.shape[0])[:, cc])[cc, :]
        aux_new = -Ls[p + 1].T @ inv_cc @ Ls[p + 1]
        idx = stages[p][2]
        xx = np.r_[off[p] + idx["q"].start:off[p] + idx["q"].stop, off[p] + idx["v"].start:off[p] + idx["v"].stop]
        A[np.ix_(xx, xx)] -= 0.5 * (aux_new + aux_new.T)
    assert np.max(np.abs(A - A.T)) < 1e-10 * np.max(np.abs(A))
    z, res = G.solve_refined(A, b)
    out = {f: np.zeros((M, {"dq": NV, "dv": NV, "da": NV, "dlmd": NV, "dgmm": NV, "dbeta": NV, "du": NU}.get(f, 12))) for f in DIRECTION}
    for p, (K, r, idx, rows, info) in enumerate(stages):
        zz = z[off[p]:off[p + 1]]
        for f, key in (("dq", "q"), ("dv", "v"), ("da", "a"), ("dlmd", "lmd"), ("dgmm", "gmm"), ("dbeta", "beta")):
            out[f][p] = zz[idx[key]]
        if not info["impulse"]:
            out["du"][p] = zz[idx["u"]]
        out["df"][p][rows], out["dmu"][p][rows] = zz[idx["f"]], zz[idx["mu"]]
        if info["switching_rows"]:
            out["dxi"][p][:info["switching_rows"]] = zz[idx["xi"]]
    return out, {"unknowns": n, "max_abs_residual": res}


def whole_iteration(spec, case, Solver, model_dict, after):
    """The iterate after `after` iterations, the next iteration's direction from the dense solve, and the oracle taken through that iteration.

    The identity behind dense_iteration holds to FIRST order in the step of the floating base: the sweeps apply their corrections to q one after the other
    (three integrateConfiguration calls on s_new.q, split_backward_correction.hxx:64-155) and read the result back with subtractConfiguration, and on SE(3)
    that is not the sum of the three tangent vectors.  On a fixed-base robot it is exact (gen_golden_kkt_parnmpc_iiwa14.py); here the comparison is made near
    the solution, where the base step is 1e-6, and the distance is recorded at both ends: it falls with the step (until, another few iterations in, the
    direction itself is rounding noise)."""
    o, qm, vm = build(spec, case, Solver)
    for _ in range(after - 1):
        assert o.update(0.0, qm, vm) == 0
    prepare(o, qm, vm)
    whole, info = dense_iteration(o, model_dict, qm)
    finish(o)
    return o, whole, info


def mismatch(o, whole, M):
    """worst over the direction's fields of  max |oracle - dense| / max |dense|  along the chain; the size of the base's step"""
    worst = 0.0
    for f in DIRECTION:
        have = o.get_chain(f, M)
        worst = max(worst, np.max(np.abs(have - whole[f])) / np.max(np.abs(whole[f])))
    return float(worst), float(np.max(np.abs(whole["dq"][:, :6])))


def prepare(o, qm, vm):
    """One full iteration first (multipliers, slacks, duals and aux matrices off their start values), then the coarse update alone with the capture on."""
    assert o.update(0.0, qm, vm) == 0
    o.lib.oracle_parnmpc_keep_uncondensed.argtypes = [C.c_void_p, C.c_int]
    o.lib.oracle_parnmpc_keep_uncondensed(o.h, 1)
    o.lib.oracle_parnmpc_phase.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    qa, va = np.ascontiguousarray(qm, dtype=np.float64), np.ascontiguousarray(vm, dtype=np.float64)
    assert o.lib.oracle_parnmpc_phase(o.h, 0, 0.0, dp(qa), dp(va)) == 0


def finish(o):
    """the rest of the iteration behind the coarse update: backward serial / parallel, forward serial / parallel (+ direction), no integration"""
    for ph in (1, 2, 3, 4):
        assert o.lib.oracle_parnmpc_phase(o.h, ph, 0.0, None, None) == 0


def model_dict():
    import helpers as H
    return RBD.load_model(H.ANYMAL_URDF, H.ANYMAL_CONTACT_FRAMES)


def main():
    import helpers as H
    spec = problem_spec()
    md = model_dict()
    out = {"_about": "tests/golden/gen_golden_kkt_parnmpc_events.py: coarse iterate of every stage of a ParNMPC horizon with events from dense solves of the stages' un-condensed KKT systems",
           "spec": spec, "cases": {}}
    for name in CASES:
        o, qm, vm = build(spec, name, H.OracleParNMPC)
        prepare(o, qm, vm)
        dense, infos = coarse_iterates(o, md, qm)
        M = len(infos)
        print(name, " ".join("%s(%d%s)" % (i["chain_kind"], i["dimf"], "+%d" % i["switching_rows"] if i["switching_rows"] else "") for i in infos))
        worst = 0.0
        for f in FIELDS:
            have = o.get_chain(f, M)
            for p in range(M):
                err = np.max(np.abs(have[p] - dense[f][p])) / max(1.0, np.max(np.abs(dense[f][p])))
                worst = max(worst, err)
                if err > 1e-9:
                    print("  %-8s position %d (%s): oracle - dense %.2e" % (f, p, infos[p]["chain_kind"], err))
        print("  worst oracle - dense over all fields and positions: %.2e   worst residual of a dense solve %.1e" % (worst, max(i["max_abs_residual"] for i in infos)))
        assert worst < 1e-8, (name, worst)
        out["cases"][name] = {"first": CASES[name][0], "events": [[s, w] for s, w in CASES[name][1]], "stages": infos,
                              "iterate": {f: dense[f].tolist() for f in FIELDS}}
    # ---- the whole iteration (coarse update + correction sweeps + expansion) against one dense solve over the horizon, near the solution
    out["whole_iteration"] = {}
    for name in WHOLE_CASES:
        wo, whole, winfo = whole_iteration(spec, name, H.OracleParNMPC, md, WHOLE_AFTER[name])
        M = len(wo.chain(0.0))
        far = whole_iteration(spec, name, H.OracleParNMPC, md, 1)
        rel_near, step_near = mismatch(wo, whole, M)
        rel_far, step_far = mismatch(far[0], far[1], M)
        winfo.update(after_iterations=WHOLE_AFTER[name], worst_relative_mismatch=rel_near, base_step=step_near,
                     one_iteration_in={"worst_relative_mismatch": rel_far, "base_step": step_far})
        print("%s: whole iteration, %d unknowns, residual of the dense solve %.1e: oracle - dense = %.1e |d| with a base step of %.1e after %d iterations, %.1e |d| with %.1e after one"
              % (name, winfo["unknowns"], winfo["max_abs_residual"], rel_near, step_near, WHOLE_AFTER[name], rel_far, step_far))
        assert rel_near < 1e-5 and rel_far < 1.0, (name, rel_near, rel_far)
        out["whole_iteration"][name] = {"info": winfo, "direction": {f: whole[f].tolist() for f in DIRECTION}}
    with open(os.path.join(HERE, "kkt_parnmpc_events.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote kkt_parnmpc_events.json")


if __name__ == "__main__":
    main()
