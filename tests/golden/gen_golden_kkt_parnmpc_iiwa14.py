#!/usr/bin/env python3
"""Golden vectors for a WHOLE ParNMPC iteration on the fixed-base arm: coarse update, the four correction sweeps and the expansion against ONE dense solve.

The backward correction (backward_correction_solver.cpp:95-366, unbackward_correction.cpp) is block back-substitution through the horizon with the pivot
of stage i taken as  K_i + aux_mat_{i+1}  -- the stage's KKT matrix with the LAST iteration's estimate of what the stages behind it add to its Hessian.  Such a
back-substitution solves a block-tridiagonal system exactly: the Newton matrix of the whole horizon with  aux_old_{i+1} - aux_new_{i+1}  added to the state block
of every stage (gen_golden_kkt_parnmpc_events.py dense_iteration, which this file runs with the arm's dimensions).  On a floating base the identity holds to first
order in the base's step only, because the sweeps compose their corrections on SE(3); on the arm the configuration is a vector and it is EXACT: the direction
of the iteration, every field of every stage, equals the dense solve to rounding.  Pins what no per-stage solve can: the correction sweeps
(split_backward_correction.hxx:84-155 / split_unbackward_correction.hxx) and how the stages talk to each other.

The un-condensed stage data comes from the oracle's ParNMPCSolver on the arm (the contact-path restatement with no passive rows, before ITS condensation
through M^-1); tests/test_golden_kkt.py holds that solver, the oracle's UnParNMPCSolver (which condenses through u = ID instead) and, with -m gpu, the HIP
UnParNMPC kernels to it at 1e-9.

Output: tests/golden/kkt_parnmpc_iiwa14.json"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import gen_golden_kkt_parnmpc_events as GE  # noqa: E402

FIELDS = ("dq", "dv", "da", "du", "dlmd", "dgmm", "dbeta")
ITERATIONS_BEFORE = 2      # the compared iteration is the third: multipliers, slacks, duals and aux matrices off their start values, the step still large


def problem_spec():
    rng = np.random.default_rng(2718)
    return {"N": 8, "T": 0.4, "q": rng.uniform(-0.6, 0.6, 7).tolist(), "v": rng.uniform(-0.5, 0.5, 7).tolist(),
            "q_meas": rng.uniform(-0.6, 0.6, 7).tolist(), "v_meas": rng.uniform(-0.5, 0.5, 7).tolist()}


def build(spec, Solver, **kw):
    import helpers as H
    m = H.iiwa14_model()
    cost, cons = H.unocp_problem(m)
    o = Solver(m, cost, cons, spec["T"], spec["N"], **kw)
    o.set_solution("q", np.array(spec["q"]))
    o.set_solution("v", np.array(spec["v"]))
    o.init(0.0)
    return o, np.array(spec["q_meas"]), np.array(spec["v_meas"])


class arm_dimensions:
    """gen_golden_kkt_parnmpc_events.py's assembly with nv = nu = 7 (restored on exit: tests import both generators)"""
    def __enter__(self):
        self.saved = (GE.NV, GE.NU, GE.NX, GE.NQ)
        GE.NV, GE.NU, GE.NX, GE.NQ = 7, 7, 14, 7

    def __exit__(self, *exc):
        GE.NV, GE.NU, GE.NX, GE.NQ = self.saved


def dense_direction(o, qm, vm):
    """capture + coarse update on the oracle's ParNMPCSolver, then the dense solve; the solver is left in front of its correction sweeps"""
    with arm_dimensions():
        GE.prepare(o, qm, vm)
        whole, info = GE.dense_iteration(o, None, qm)
    return {f: whole[f] for f in FIELDS}, info


def main():
    import helpers as H
    spec = problem_spec()
    o, qm, vm = build(spec, H.OracleParNMPC)
    for _ in range(ITERATIONS_BEFORE - 1):
        assert o.update(0.0, qm, vm) == 0
    dense, info = dense_direction(o, qm, vm)               # (prepare() runs one more full iteration, then the coarse update of the compared one)
    GE.finish(o)
    worst = 0.0
    for f in FIELDS:
        have = o.get(f)
        err = np.max(np.abs(have - dense[f])) / max(1.0, np.max(np.abs(dense[f])))
        worst = max(worst, err)
        print("  %-6s max |dense| %.3e   oracle ParNMPCSolver - dense %.2e" % (f, np.max(np.abs(dense[f])), err))
    print(info)
    assert worst < 1e-9, worst
    out = {"_about": "tests/golden/gen_golden_kkt_parnmpc_iiwa14.py: direction of a whole ParNMPC iteration on iiwa14 from one dense solve of the horizon's modified Newton system",
           "spec": spec, "iterations_before": ITERATIONS_BEFORE, "dense_system": info, "direction": {f: dense[f].tolist() for f in FIELDS}}
    with open(os.path.join(HERE, "kkt_parnmpc_iiwa14.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote kkt_parnmpc_iiwa14.json")


if __name__ == "__main__":
    main()
