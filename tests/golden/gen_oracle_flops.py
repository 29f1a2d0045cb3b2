#!/usr/bin/env python3
"""Exact FLOP counts of one SQP iteration of every bench workload ("the build's oracle must carry an exact FLOP
counter and report its number next to these estimates").

Builds oracle/liboracle_flops.so -- the oracle's own sources with the scalar replaced by a counting double (oracle/flops.hpp) -- and runs
ONE updateSolution of instance 0 of each workload of bench.py from the bench's initial iterate.  Every +, -, *, /, sqrt and
transcendental call of the restatement is counted and attributed to the region of the hot path it belongs to (rows of SURVEY 8(a)).
What is counted is the reference's FORMULATION on dense blocks (a dense product counts its structural zeros, like Eigen's GEMM);
a fused multiply-add is one mul + one add.

Output: tests/golden/oracle_flops.json (data only).  bench.py quotes it in `roofline.flops`; tests/test_oracle_flops.py recounts the
small workloads and holds the file to the live count."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# which regions each kernel of the product executes; used by bench.py to price a kernel in FLOPs
KERNEL_REGIONS = {
    "ocp_nominal+ocp_rnea(switch)": ["kinematics", "rnea", "lie", "switching_constraint"],
    "ocp_condense": ["rnea_derivatives", "baumgarte_contact", "mjtjinv", "cost_constraints_multipliers", "condense"],
    "ocp_riccati_backward": ["riccati_backward"],
    "ocp_forward_expand": ["riccati_forward", "expand_direction"],
    "ocp_expand_dual_integrate": ["integrate"],
    "un_linearize": ["kinematics", "rnea", "rnea_derivatives", "cost_constraints_multipliers", "unconstrained_dynamics"],
    "un_riccati_backward": ["riccati_backward"],
    "un_riccati_forward": ["riccati_forward"],
    "un_expand": ["expand_direction"],
    "un_integrate": ["integrate"],
    "parnmpc_kkt_inverse": ["parnmpc_kkt_inverse"],
    "parnmpc_corrections": ["parnmpc_corrections"],
}


def flops_lib():
    import helpers
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir, "liboracle_flops.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cannot build the counting oracle:\n" + r.stderr)
    helpers.ORACLE_PATH_OVERRIDE = os.path.join(odir, "liboracle_flops.so")
    helpers._oracles.pop(False, None)
    lib = helpers.oracle()
    assert lib.oracle_flops_enabled() == 1
    lib.oracle_flops_region_name.restype = C.c_char_p
    lib.oracle_flops_kind_name.restype = C.c_char_p
    lib.oracle_flops_region_name.argtypes = [C.c_int]
    lib.oracle_flops_kind_name.argtypes = [C.c_int]
    return lib


def release_flops_lib():
    import helpers
    helpers.ORACLE_PATH_OVERRIDE = None
    helpers._oracles.pop(False, None)


def read_counts(lib):
    nr, nk = C.c_int(), C.c_int()
    lib.oracle_flops_shape(C.byref(nr), C.byref(nk))
    buf = (C.c_ulonglong * (nr.value * nk.value))()
    lib.oracle_flops_get(buf)
    out = {}
    for i in range(nr.value):
        row = {lib.oracle_flops_kind_name(k).decode(): int(buf[i * nk.value + k]) for k in range(nk.value)}
        if any(row.values()):
            out[lib.oracle_flops_region_name(i).decode()] = row
    return out


def make_solver(workload, N=None):
    """(oracle solver at the bench's initial iterate, q, v, chain length, description) -- the set-up of bench.py / bench.cpu_baseline"""
    import helpers as H
    if workload in ("iiwa14",):
        N = N or 100
        m = H.iiwa14_model()
        cost, cons = H.unocp_problem(m)
        q, v = np.full(m.nv, 2.0), np.zeros(m.nv)
        o = H.OracleUnOCP(m, cost, cons, 0.05 * N, N)
        o.set_solution("q", q); o.set_solution("v", v)
        return o, q, v, N + 1, "iiwa14 UnOCPSolver N=%d" % N
    m = H.anymal_model()
    fz = [0, 0, 0.25 * (-m.total_mass * m.gravity[2])]
    q, v = H.ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    if workload in ("anymal", "anymal_half"):
        N = N or 100
        cost, cons = H.anymal_problem(m, trotting_ref=True)
        o = H.OracleOCP(m, cost, cons, 0.05 * N, N)
        o.set_contact_status([1, 1, 1, 1] if workload == "anymal" else [0, 1, 1, 0], H.anymal_contact_points(m))
        o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
        o.init_constraints(0.0)
        return o, q, v, N + 1, "ANYmal OCPSolver N=%d, 4 point contacts on every stage" % N
    if workload == "anymal_trotting":
        N = N or 100
        nimp = 9 if N == 100 else max(1, int((0.05 * N - 0.55) / 0.5))
        T = 0.5 + nimp * 0.5 + 0.05 if N == 100 else 0.05 * N
        cost, cons = H.anymal_problem(m, trotting_ref=True)
        o = H.OracleOCP(m, cost, cons, T, N, max_num_impulse=nimp + 1)
        H.trotting_sequence(o, m, nimp)
        o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
        o.init_constraints(0.0)
        return o, q, v, N + 1 + 2 * nimp + 1, "ANYmal OCPSolver N=%d, trotting sequence (1 lift + %d impulse events)" % (N, nimp)
    if workload == "anymal_running":
        N = N or 200
        cost, cons = H.running_problem(m, 10)
        q = H.ANYMAL_Q_RUNNING_START.copy()
        o = H.OracleOCP(m, cost, cons, 7.0 * N / 240, N, max_num_impulse=26)
        H.running_sequence(o, m, 10)
        o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
        o.init_constraints(0.0)
        return o, q, v, N + 1 + 2 * 26 + 14, "ANYmal OCPSolver N=%d, running sequence (26 impulse + 14 lift events)" % N
    if workload == "anymal_parnmpc":
        N = N or 256
        cost, cons = H.anymal_problem(m, trotting_ref=True)
        o = H.OracleParNMPC(m, cost, cons, 0.05 * N, N)
        o.set_contact_status([1, 1, 1, 1], H.anymal_contact_points(m))
        o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
        o.init(0.0)
        return o, q, v, N, "ANYmal ParNMPCSolver N=%d, 4 point contacts on every stage" % N
    raise ValueError(workload)


def count(workload, N=None, lib=None):
    own = lib is None
    if own:
        lib = flops_lib()
    try:
        o, q, v, chain, desc = make_solver(workload, N)
        lib.oracle_flops_reset()
        assert o.update(0.0, q, v) == 0
        regions = read_counts(lib)
        del o
    finally:
        if own:
            release_flops_lib()
    total = {k: sum(r[k] for n, r in regions.items() if n != "other") for k in ("add", "mul", "div", "sqrt", "transcendental")}
    flop = sum(total.values())
    return {"workload": desc, "chain_stages": chain, "regions": regions, "total_by_kind": total, "flop_per_iteration": flop,
            "flop_per_stage": flop / chain}


def main():
    lib = flops_lib()
    out = {"_about": "exact operation counts of ONE SQP iteration of the CPU restatement (oracle/flops.hpp, tests/golden/gen_oracle_flops.py); "
                     "flop = add + mul + div + sqrt + transcendental, a fused multiply-add counts twice; region 'other' (discretiser, set-up) excluded",
           "kernel_regions": KERNEL_REGIONS, "workloads": {}}
    for wl in ("iiwa14", "anymal", "anymal_trotting", "anymal_running", "anymal_parnmpc"):
        out["workloads"][wl] = count(wl, lib=lib)
        w = out["workloads"][wl]
        print("%-16s %3d stages  %.4g FLOP / iteration  %.4g FLOP / stage" % (wl, w["chain_stages"], w["flop_per_iteration"], w["flop_per_stage"]))
        for n, r in w["regions"].items():
            print("    %-30s %12d" % (n, sum(r.values())))
    # one regular stage of a class, exactly: the difference of two horizons of the uniform problem (the terminal stage and the set-up drop out)
    def per_stage(wl):
        a, b = count(wl, 4, lib=lib), count(wl, 8, lib=lib)
        reg = {n: sum(r.values()) - sum(a["regions"].get(n, {}).values()) for n, r in b["regions"].items() if n != "other"}
        return {n: v / 4 for n, v in reg.items()}
    out["per_stage_by_class"] = {"anymal_nf12": per_stage("anymal"), "anymal_nf6": per_stage("anymal_half"), "iiwa14": per_stage("iiwa14")}
    for k, v in out["per_stage_by_class"].items():
        print(k, {n: int(x) for n, x in v.items()}, "sum", int(sum(v.values())))
    # small cases the CPU test recounts
    out["workloads"]["iiwa14_N20"] = count("iiwa14", 20, lib=lib)
    out["workloads"]["anymal_N8"] = count("anymal", 8, lib=lib)
    release_flops_lib()
    with open(os.path.join(HERE, "oracle_flops.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
