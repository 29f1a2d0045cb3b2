"""FP32 tolerance study of the Riccati half of the hot path (BASELINE.json configs[4]; run by hand, not a pytest file):

    python tests/study_fp32_riccati.py

The stage LQR blocks (Qxx, Qxu, Quu, A, B, lx, lu, Fx after condensation) of the running-gait cost on an N = 200, T = 7 horizon
(dt = 35 ms, all feet in contact: the event-free core of examples/anymal/anymal_running.cpp) are taken from the CPU oracle at
several SQP iterates; the backward / forward Riccati recursion (backward_riccati_recursion_factorizer.hxx:44-161,
split_riccati_factorizer.hxx:36-128, riccati_recursion_solver.cpp:48-251) is then re-run in numpy in four arithmetic variants
and the Newton direction (dx, du) of each is compared with the all-FP64 one:

    f64      reference
    in32     stage blocks rounded to FP32 (what an FP32 kkt record would hold), recursion in FP64
    P32      FP64 blocks, P and s rounded to FP32 after every stage (an FP32 ric record / FP32 LDS copy of P)
    f32      everything in FP32

Output: max relative error of dx and du over the horizon, per iterate."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import ANYMAL_Q_RUNNING_START, OracleOCP, anymal_contact_points, anymal_model, running_problem  # noqa: E402


def riccati(lqr, PN, sN, dx0, dt_in, dt_P, dt_all):
    """backward + forward sweep; dt_in: dtype the stage blocks are rounded to, dt_P: dtype P, s are stored in, dt_all: arithmetic"""
    N = len(lqr)
    cast = lambda a, t: np.asarray(a, dtype=t).astype(dt_all)
    P, s = [None] * (N + 1), [None] * (N + 1)
    P[N], s[N] = cast(cast(PN, dt_in), dt_P), cast(cast(sN, dt_in), dt_P)
    K, k = [None] * N, [None] * N
    blocks = [tuple(cast(b, dt_in) for b in st) for st in lqr]
    for i in range(N - 1, -1, -1):
        Qxx, Qxu, Quu, A, B, lx, lu, Fx = blocks[i]
        Pn, sn = P[i + 1], s[i + 1]
        AtP = A.T @ Pn
        F = Qxx + AtP @ A
        H = Qxu + AtP @ B
        G = Quu + B.T @ Pn @ B
        Ginv = np.linalg.inv(G)
        K[i] = -(Ginv @ H.T)
        k[i] = -(Ginv @ (lu + B.T @ (Pn @ Fx - sn)))
        Pi = F - K[i].T @ G @ K[i]
        Pi = (0.5 * (Pi + Pi.T)).astype(dt_all)
        si = A.T @ (sn - Pn @ Fx) - lx - H @ k[i]
        P[i], s[i] = cast(Pi, dt_P), cast(si, dt_P)
    dx = [None] * (N + 1)
    du = [None] * N
    dx[0] = np.asarray(dx0, dtype=dt_all)
    for i in range(N):
        _, _, _, A, B, _, _, Fx = blocks[i]
        du[i] = K[i] @ dx[i] + k[i]
        dx[i + 1] = A @ dx[i] + B @ du[i] + Fx
    return np.array(dx, dtype=np.float64), np.array(du, dtype=np.float64)


def main():
    m = anymal_model()
    cost, cons = running_problem(m, 10)
    N, T = 200, 7.0
    o = OracleOCP(m, cost, cons, T, N)
    q, v = ANYMAL_Q_RUNNING_START.copy(), np.zeros(m.nv)
    q_meas = q.copy()
    q_meas[7:] += 0.05                      # a measured state off the initial guess: dx0 != 0
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m, ANYMAL_Q_RUNNING_START))
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    print("iterate  KKT error   variant   max|ddx|/max|dx|   max|ddu|/max|du|")
    for it in range(41):
        if it % 8:
            assert o.update(0.0, q_meas, v) == 0
            continue
        err = o.kkt_error(0.0, q_meas, v)
        assert o.stage(0, 0.0, q_meas, v) == 0                # linearise + condense only
        lqr = [o.lqr_stage(i) for i in range(N)]
        assert o.stage(1, 0.0, q_meas, v) == 0                # the oracle's own backward sweep (terminal P, s)
        P, s, _, _ = o.riccati()
        assert o.stage(2, 0.0, q_meas, v) == 0
        dq0, dv0 = o.get("dq")[0], o.get("dv")[0]
        dx0 = np.concatenate([dq0, dv0])
        ref = riccati(lqr, P[N], s[N], dx0, np.float64, np.float64, np.float64)
        # the numpy restatement reproduces the oracle's direction
        assert np.abs(ref[1] - o.get("du")).max() < 1e-7 * max(1.0, np.abs(ref[1]).max())
        for name, (a, b, c) in (("in32", (np.float32, np.float64, np.float64)), ("P32", (np.float64, np.float32, np.float64)),
                                ("f32", (np.float32, np.float32, np.float32))):
            dx, du = riccati(lqr, P[N], s[N], dx0, a, b, c)
            ex = np.abs(dx - ref[0]).max() / np.abs(ref[0]).max()
            eu = np.abs(du - ref[1]).max() / np.abs(ref[1]).max()
            print("%4d    %9.3e   %-6s    %10.2e         %10.2e" % (it, err, name, ex, eu))
        assert o.update(0.0, q_meas, v) == 0


if __name__ == "__main__":
    main()
