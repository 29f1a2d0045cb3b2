#!/usr/bin/env python3
"""Generate the golden vectors of the RICCATI layer (run in the build container: numpy only).

The parity oracle's backward Riccati recursion (oracle/ocp.cpp, OCPSolver::backwardRiccatiRecursion, restating
include/idocp/ocp/backward_riccati_recursion_factorizer.hxx:44-161 and split_riccati_factorizer.hxx:24-52) is pinned by the tests of
tests/test_oracle_ocp.py through the stage-wise formulas -- which it also implements.  This generator shares no formula with it: for a
random, well-conditioned linear-quadratic problem in the reference's block structure

    min  sum_i  1/2 [x_i; u_i]^T [Qxx Qxu; Qxu^T Quu]_i [x_i; u_i] + lx_i^T x_i + lu_i^T u_i   +   1/2 x_N^T Qf x_N + lf^T x_N
    s.t. x_{i+1} = A_i x_i + B_i u_i + Fx_i,     A = [Fqq Fqv; Fvq Fvv],  B = [0; Fvu],
         Fqq = diag(Fqq6, I), Fqv = diag(Fqv6, dt I)          (floating base: only the leading 6 x 6 blocks are stored)

it assembles the DENSE KKT system of every tail problem (stages i .. N) and reads the Riccati quantities off its solution by the
definition of the cost-to-go:  with x_i = e_j (and x_i = 0) prescribed,  u_i = K_i x_i + k_i  and the multiplier of the initial
condition is  lambda_i = P_i x_i - s_i   (the reference's sign convention, split_riccati_factorizer.hxx:131-139).
(test/ocp/riccati_recursion_solver_test.cpp of the reference checks its recursion against exactly such a dense solve.)

Output: tests/golden/riccati_lqr.json -- the stage data and P_i, s_i, K_i, k_i of every stage.  Only data is written."""
import json
import os

import numpy as np

NV, NU, N = 18, 12, 3
NX = 2 * NV
HERE = os.path.dirname(os.path.abspath(__file__))


def spd(rng, n, lo):
    a = rng.standard_normal((n, n))
    return a @ a.T / n + lo * np.eye(n)


def make_problem(seed=20250):
    rng = np.random.default_rng(seed)
    dt = 0.05
    stages = []
    for i in range(N):
        Qz = spd(rng, NX + NU, 0.5)
        Fqq6 = np.eye(6) + 0.05 * rng.standard_normal((6, 6))
        Fqv6 = dt * np.eye(6) + 0.01 * rng.standard_normal((6, 6))
        st = dict(Qxx=Qz[:NX, :NX], Qxu=Qz[:NX, NX:], Quu=Qz[NX:, NX:], Fqq6=Fqq6, Fqv6=Fqv6,
                  Fvq=0.1 * rng.standard_normal((NV, NV)), Fvv=np.eye(NV) + 0.1 * rng.standard_normal((NV, NV)),
                  Fvu=0.3 * rng.standard_normal((NV, NU)), lx=rng.standard_normal(NX), lu=rng.standard_normal(NU),
                  Fx=0.1 * rng.standard_normal(NX), dt=dt)
        stages.append(st)
    term = dict(Qxx=spd(rng, NX, 0.5), lx=rng.standard_normal(NX))
    # the oracle's terminal P ignores the q-v coupling of the terminal Hessian (riccati_recursion_solver.cpp:53-56 takes Qqq and Qvv):
    # the golden problem has none
    term["Qxx"][:NV, NV:] = 0.0
    term["Qxx"][NV:, :NV] = 0.0
    return stages, term


def AB(st):
    Fqq = np.eye(NV); Fqq[:6, :6] = st["Fqq6"]
    Fqv = st["dt"] * np.eye(NV); Fqv[:6, :6] = st["Fqv6"]
    A = np.block([[Fqq, Fqv], [st["Fvq"], st["Fvv"]]])
    B = np.vstack([np.zeros((NV, NU)), st["Fvu"]])
    return A, B


def tail_solution(stages, term, i0, x0):
    """Dense KKT of the tail problem starting at stage i0 with x_{i0} = x0.  Unknowns: z = (x_{i0}, u_{i0}, ..., x_N), multipliers
    lam_{i0} (initial condition) and lam_{i+1} (dynamics).  Returns (u_{i0}, multiplier of the initial condition)."""
    n = N - i0
    nz = n * (NX + NU) + NX
    nc = (n + 1) * NX
    H = np.zeros((nz, nz)); g = np.zeros(nz)
    C = np.zeros((nc, nz)); d = np.zeros(nc)
    xo = lambda k: k * (NX + NU)
    uo = lambda k: k * (NX + NU) + NX
    for k in range(n):
        st = stages[i0 + k]
        H[xo(k):xo(k) + NX, xo(k):xo(k) + NX] = st["Qxx"]
        H[xo(k):xo(k) + NX, uo(k):uo(k) + NU] = st["Qxu"]
        H[uo(k):uo(k) + NU, xo(k):xo(k) + NX] = st["Qxu"].T
        H[uo(k):uo(k) + NU, uo(k):uo(k) + NU] = st["Quu"]
        g[xo(k):xo(k) + NX] = st["lx"]; g[uo(k):uo(k) + NU] = st["lu"]
        A, B = AB(st)
        r = (k + 1) * NX                      # x_{k+1} - A x_k - B u_k = Fx
        C[r:r + NX, xo(k + 1):xo(k + 1) + NX] = np.eye(NX)
        C[r:r + NX, xo(k):xo(k) + NX] = -A
        C[r:r + NX, uo(k):uo(k) + NU] = -B
        d[r:r + NX] = st["Fx"]
    H[xo(n):xo(n) + NX, xo(n):xo(n) + NX] = term["Qxx"]
    g[xo(n):xo(n) + NX] = term["lx"]
    C[0:NX, xo(0):xo(0) + NX] = np.eye(NX); d[0:NX] = x0
    # Lagrangian  f(z) + mu^T (d - C z):  stationarity H z + g - C^T mu = 0
    K = np.block([[H, -C.T], [C, np.zeros((nc, nc))]])
    sol = np.linalg.solve(K, np.concatenate([-g, d]))
    z, mu = sol[:nz], sol[nz:]
    # df*/dx0 = mu_0 = P x0 - s   (value function V(x0) = 1/2 x0^T P x0 - s^T x0 + const)
    return z[uo(0):uo(0) + NU] if n > 0 else None, mu[:NX]


def main():
    stages, term = make_problem()
    out = {"nv": NV, "nu": NU, "N": N, "stages": [], "terminal": {k: v.tolist() for k, v in term.items()}, "riccati": []}
    for st in stages:
        out["stages"].append({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items()})
    for i in range(N + 1):
        u0, l0 = tail_solution(stages, term, i, np.zeros(NX))
        P = np.zeros((NX, NX)); K = np.zeros((NU, NX))
        for j in range(NX):
            e = np.zeros(NX); e[j] = 1.0
            uj, lj = tail_solution(stages, term, i, e)
            P[:, j] = lj - l0
            if uj is not None:
                K[:, j] = uj - u0
        rec = {"P": P.tolist(), "s": (-l0).tolist()}
        if u0 is not None:
            rec["K"] = K.tolist(); rec["k"] = u0.tolist()
        out["riccati"].append(rec)
        assert np.abs(P - P.T).max() < 1e-9 * np.abs(P).max()
    with open(os.path.join(HERE, "riccati_lqr.json"), "w") as f:
        json.dump(out, f)
    print("wrote riccati_lqr.json:", os.path.getsize(os.path.join(HERE, "riccati_lqr.json")), "bytes")


if __name__ == "__main__":
    main()
