"""FP32 tolerance study of the WHOLE hot path (BASELINE.json configs[4]: "ANYmal running ... OCPSolver N = 200 ... FP32 tolerance study";
run by hand, not a pytest file):

    python tests/study_fp32_pipeline.py [N]

`make -C oracle liboracle_f32.so` builds the CPU restatement with a FLOAT scalar: rigid-body derivatives, contact-dynamics condensation,
backward / forward Riccati sweep and the expansion of the direction all in single precision (IEEE binary32, no mixed accumulation) -- what a
straight FP32 port of every kernel would compute.  The Newton direction of the running-gait problem (examples/anymal/anymal_running.cpp
rescaled to N = 200, 267 stages in the chain) is compared with the FP64 restatement (a) for ONE iteration from the same start iterate, field by
field, and (b) over the SQP iteration run entirely in each arithmetic: the KKT error each build reaches.

Companions: tests/study_fp32_riccati.py (round 2: which HALF tolerates what -- FP32 storage of P, s is fine at 1e-6, FP32 stage blocks or an
FP32 sweep are not) and tests/test_hybrid_gpu.py::test_configs4_grid_and_fp32_storage (the storage variant measured on the device)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers as H      # noqa: E402

FIELDS = ("dq", "dv", "da", "du", "df", "dlmd", "dgmm", "dbeta", "dmu")


def make(N, lib_path=None):
    H.ORACLE_PATH_OVERRIDE = lib_path
    H._oracles.pop(False, None)
    m = H.anymal_model()
    cost, cons = H.running_problem(m, 10)
    o = H.OracleOCP(m, cost, cons, 7.0 * N / 240, N, max_num_impulse=26)
    H.running_sequence(o, m, 10)
    q = H.ANYMAL_Q_RUNNING_START.copy()
    o.set_solution("q", q)
    o.set_solution("v", np.zeros(m.nv))
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init_constraints(0.0)
    return m, o, q, np.zeros(m.nv)


def direction(o, M):
    return {f: o.get_chain(f, M) for f in FIELDS}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir, "liboracle_f32.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f32 = os.path.join(odir, "liboracle_f32.so")
    # (a) one iteration from the same iterate (the start iterate of the bench workload): direction of the FP32 build against the FP64 build
    dirs, Ms = [], 0
    for path in (None, f32):
        m, o, q, v = make(N, path)
        Ms = len(o.chain(0.0))
        assert o.update(0.0, q, v) == 0
        dirs.append(direction(o, Ms))
        del o
    print("ANYmal running gait, N = %d, %d stages in the chain: first Newton direction, FP32 restatement against FP64" % (N, Ms))
    print("(max over the chain of |d32 - d64| / max(1, largest entry of the stage's field))")
    for f in FIELDS:
        a, b = dirs[0][f], dirs[1][f]
        per = [np.max(np.abs(a[p] - b[p])) / max(1.0, np.max(np.abs(a[p]))) for p in range(Ms)]
        print("  %-6s worst %.1e at chain position %3d   median over the chain %.1e" % (f, max(per), int(np.argmax(per)), float(np.median(per))))
    # (b) the SQP iteration run entirely in FP32 against the FP64 one: KKT error per iteration (each build measures its own)
    print("\nKKT error of the SQP iteration, every build with its own arithmetic:")
    print("  it      FP64          FP32")
    runs = []
    for path in (None, f32):
        m, o, q, v = make(N, path)
        errs = [o.kkt_error(0.0, q, v)]
        for _ in range(12):
            rc = o.update(0.0, q, v)
            e = o.kkt_error(0.0, q, v)
            errs.append(e if rc == 0 else float("nan"))
        runs.append(errs)
        del o
    H.ORACLE_PATH_OVERRIDE = None
    H._oracles.pop(False, None)
    for i, (a, b) in enumerate(zip(*runs)):
        print("  %2d   %.3e     %.3e" % (i, a, b))


if __name__ == "__main__":
    main()
