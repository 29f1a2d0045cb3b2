"""Shared test plumbing: oracle loader (ctypes), workload definitions, thin
wrappers over the product C ABI.  The oracle is the CHECKER only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

from idocp_amd import capi
from idocp_amd.workloads import *      # noqa: F401,F403  (models, problems, sequences, Hip* wrappers: shared with bench.py)
from idocp_amd.workloads import P, arr, dp, ROOT, GOLDEN


_oracles = {}


def force_forms(monkeypatch, fused=None, sweep=None):
    """Every OCPSolver handle created during this test runs the forward sweep (fused: 1 = ocp_forward_expand_kernel, 0 = S4 + K6) and / or the
    backward sweep (sweep: 0 = one wavefront per instance, register-resident; 1 = eight per instance) in the given form instead of the one its
    batch size would pick -- idocp_ocp_set_fused_forward / idocp_ocp_set_riccati_sweep on the new handle.  Scoped to the test by monkeypatch."""
    def hook(solver):
        lib = solver.lib
        if fused is not None:
            lib.idocp_ocp_set_fused_forward.argtypes = [C.c_void_p, C.c_int]
            capi.check(lib.idocp_ocp_set_fused_forward(solver.h, int(fused)), "set_fused_forward")
        if sweep is not None:
            lib.idocp_ocp_set_riccati_sweep.argtypes = [C.c_void_p, C.c_int]
            capi.check(lib.idocp_ocp_set_riccati_sweep(solver.h, int(sweep)), "set_riccati_sweep")
    monkeypatch.setattr(HipOCP, "_post_create", list(HipOCP._post_create) + [hook])

ORACLE_PATH_OVERRIDE = None       # bench.py's cpu_baseline leg points this at the natively built library

def oracle(hp=False):
    """liboracle.so, or with hp=True liboracle_hp.so -- the SAME restatement built with a long double scalar, the referee
    of the parity tests on ill-conditioned problems; (re)built with make if missing or stale."""
    global _oracles
    _oracle = _oracles.get(hp)
    if _oracle is None:
        path = os.path.join(ROOT, "oracle", "liboracle_hp.so" if hp else "liboracle.so")
        if ORACLE_PATH_OVERRIDE and not hp:
            path = ORACLE_PATH_OVERRIDE
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], capture_output=True, text=True)
        if r.returncode != 0 and not os.path.exists(path):
            raise RuntimeError("cannot build the oracle:\n" + r.stderr)
        lib = C.CDLL(path)
        vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
        PM = C.POINTER(capi.Model)
        lib.oracle_rnea.argtypes = [PM, dp, dp, dp, dp, ci, dp]
        lib.oracle_rnea_derivatives.argtypes = [PM, dp, dp, dp, dp, ci, dp, dp, dp]
        lib.oracle_contact_kinematics.argtypes = [PM, dp, dp, dp, dp, cd] + [dp] * 13
        lib.oracle_lie_ops.argtypes = [PM] + [dp] * 7
        lib.oracle_unocp_create.argtypes = [PM, C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci]
        lib.oracle_unocp_create.restype = vp
        lib.oracle_unocp_destroy.argtypes = [vp]
        lib.oracle_unocp_set_solution.argtypes = [vp, cs, dp]
        lib.oracle_unocp_init_constraints.argtypes = [vp]
        lib.oracle_unocp_update_solution.argtypes = [vp, cd, dp, dp]
        lib.oracle_unocp_stage.argtypes = [vp, ci, cd, dp, dp]
        lib.oracle_unocp_compute_kkt_residual.argtypes = [vp, cd, dp, dp]
        lib.oracle_unocp_kkt_error.argtypes = [vp]
        lib.oracle_unocp_kkt_error.restype = cd
        lib.oracle_unocp_is_current_solution_feasible.argtypes = [vp]
        lib.oracle_unocp_get_solution.argtypes = [vp, cs, dp]
        lib.oracle_unocp_get_direction.argtypes = [vp, cs, dp]
        lib.oracle_unocp_get_step_sizes.argtypes = [vp, dp, dp]
        lib.oracle_unocp_get_riccati.argtypes = [vp, dp, dp, dp, dp]
        lib.oracle_unocp_dimc.argtypes = [vp]
        lib.oracle_unocp_get_constraint_data.argtypes = [vp, dp, dp]
        lib.oracle_unocp_get_unkkt.argtypes = [vp, dp, dp]
        lib.oracle_unocp_bench.argtypes = [vp, cd, dp, dp, ci, dp]
        lib.oracle_unocp_bench.restype = cd
        lib.oracle_unocp_set_num_threads.argtypes = [vp, ci]
        lib.oracle_openmp_enabled.restype = ci
        lib.oracle_unparnmpc_create.argtypes = [PM, C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci]
        lib.oracle_unparnmpc_create.restype = vp
        lib.oracle_unparnmpc_destroy.argtypes = [vp]
        lib.oracle_unparnmpc_set_solution.argtypes = [vp, cs, dp]
        lib.oracle_unparnmpc_init.argtypes = [vp, cd]
        lib.oracle_unparnmpc_update_solution.argtypes = [vp, cd, dp, dp]
        lib.oracle_unparnmpc_stage.argtypes = [vp, ci, cd, dp, dp]
        lib.oracle_unparnmpc_kkt_error.argtypes = [vp, cd, dp, dp]
        lib.oracle_unparnmpc_kkt_error.restype = cd
        lib.oracle_unparnmpc_is_current_solution_feasible.argtypes = [vp]
        lib.oracle_unparnmpc_get.argtypes = [vp, cs, dp]
        lib.oracle_unparnmpc_get_step_sizes.argtypes = [vp, dp, dp]
        lib.oracle_unparnmpc_get_matrices.argtypes = [vp, dp, dp]
        lib.oracle_unparnmpc_get_constraint_data.argtypes = [vp, dp, dp]
        lib.oracle_unparnmpc_bench.argtypes = [vp, cd, dp, dp, ci, dp]
        lib.oracle_unparnmpc_bench.restype = cd
        lib.oracle_unparnmpc_set_slice.argtypes = [vp, ci, ci]
        lib.oracle_unparnmpc_export.argtypes = [vp, ci, dp]
        lib.oracle_unparnmpc_import.argtypes = [vp, ci, dp]
        lib.oracle_unparnmpc_set_step_sizes.argtypes = [vp, cd, cd]
        lib.oracle_unparnmpc_kkt_error_squared.argtypes = [vp, cd, dp, dp]
        lib.oracle_unparnmpc_kkt_error_squared.restype = cd
        lib.oracle_unocp_update_solution_ls.argtypes = [vp, cd, dp, dp]
        lib.oracle_unocp_clear_line_search_filter.argtypes = [vp]
        lib.oracle_unocp_cost_and_violation.argtypes = [vp, cd, dp]
        lib.oracle_unparnmpc_update_solution_ls.argtypes = [vp, cd, dp, dp]
        lib.oracle_unparnmpc_clear_line_search_filter.argtypes = [vp]
        lib.oracle_unparnmpc_cost_and_violation.argtypes = [vp, cd, dp, dp, dp]
        _oracle = _oracles[hp] = lib
    return _oracle

def load_golden(robot):
    with open(os.path.join(GOLDEN, "rbd_%s.json" % robot)) as f:
        return json.load(f)

class OracleUnOCP:
    def __init__(self, model, cost, cons, T, N, hp=False):
        self.lib = oracle(hp)
        self.N, self.nv = N, model.nv
        self.h = self.lib.oracle_unocp_create(C.byref(model), C.byref(cost), C.byref(cons), T, N)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_unocp_destroy(self.h)
            self.h = None

    def set_solution(self, name, value):
        assert self.lib.oracle_unocp_set_solution(self.h, name.encode(), P(arr(value))) == 0

    def update(self, t, q, v, line_search=False):
        f = self.lib.oracle_unocp_update_solution_ls if line_search else self.lib.oracle_unocp_update_solution
        return f(self.h, t, P(arr(q)), P(arr(v)))

    def clear_line_search_filter(self):
        self.lib.oracle_unocp_clear_line_search_filter(self.h)

    def cost_and_violation(self, alpha):
        out = np.zeros(2)
        self.lib.oracle_unocp_cost_and_violation(self.h, alpha, P(out))
        return out

    def set_task_refs(self, refs):
        """TaskSpace*Cost references of stages 0 .. N, [N + 1][12] (rotation row-major, position)"""
        self.lib.oracle_unocp_set_task_refs.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        assert self.lib.oracle_unocp_set_task_refs(self.h, P(arr(refs))) == 0

    def task_terms(self, stage, q):
        """(cost without dt, gradient, Gauss-Newton Hessian) of the task-space term of a stage at configuration q"""
        self.lib.oracle_unocp_task_terms.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_double)] * 4
        c, g, H = np.zeros(1), np.zeros(self.nv), np.zeros((self.nv, self.nv))
        assert self.lib.oracle_unocp_task_terms(self.h, stage, P(arr(q)), P(c), P(g), P(H)) == 0
        return c[0], g, H

    def stage(self, what, t, q, v):
        return self.lib.oracle_unocp_stage(self.h, what, t, P(arr(q)), P(arr(v)))

    def infeasible_stage(self):
        """first stage violating an inequality constraint, -1 if the iterate is feasible"""
        return self.lib.oracle_unocp_is_current_solution_feasible(self.h)

    def kkt_error(self, t, q, v):
        self.lib.oracle_unocp_compute_kkt_residual(self.h, t, P(arr(q)), P(arr(v)))
        return self.lib.oracle_unocp_kkt_error(self.h)

    def solution(self, name):
        out = np.zeros((self.N + 1, self.nv))
        assert self.lib.oracle_unocp_get_solution(self.h, name.encode(), P(out)) == 0
        return out[:stages_of(name, self.N)]

    def direction(self, name):
        out = np.zeros((self.N + 1, self.nv))
        assert self.lib.oracle_unocp_get_direction(self.h, name.encode(), P(out)) == 0
        return out[:stages_of(name, self.N)]

    def step_sizes(self):
        a, b = C.c_double(), C.c_double()
        self.lib.oracle_unocp_get_step_sizes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def riccati(self):
        nv, N = self.nv, self.N
        Pm, s = np.zeros((N + 1, 2 * nv, 2 * nv)), np.zeros((N + 1, 2 * nv))
        K, k = np.zeros((N, 2 * nv, nv)), np.zeros((N, nv))
        self.lib.oracle_unocp_get_riccati(self.h, P(Pm), P(s), P(K), P(k))
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k      # col-major -> [row, col]

    def constraint_data(self):
        dimc = self.lib.oracle_unocp_dimc(self.h)
        sl, du = np.zeros((self.N, dimc)), np.zeros((self.N, dimc))
        self.lib.oracle_unocp_get_constraint_data(self.h, P(sl), P(du))
        return sl, du

    def unkkt(self):
        nv, N = self.nv, self.N
        Q, r = np.zeros((N, 3 * nv, 3 * nv)), np.zeros((N, 5 * nv))
        self.lib.oracle_unocp_get_unkkt(self.h, P(Q), P(r))
        return Q.transpose(0, 2, 1), r

class OracleUnParNMPC:
    """oracle::UnParNMPCSolver: N backward-Euler stages, every field is [N][nv]"""

    def __init__(self, model, cost, cons, T, N, hp=False):
        self.lib = oracle(hp)
        self.N, self.nv = N, model.nv
        self.h = self.lib.oracle_unparnmpc_create(C.byref(model), C.byref(cost), C.byref(cons), T, N)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_unparnmpc_destroy(self.h)
            self.h = None

    def set_solution(self, name, value):
        assert self.lib.oracle_unparnmpc_set_solution(self.h, name.encode(), P(arr(value))) == 0

    def init(self, t=0.0):
        self.lib.oracle_unparnmpc_init(self.h, t)

    def update(self, t, q, v, line_search=False):
        f = self.lib.oracle_unparnmpc_update_solution_ls if line_search else self.lib.oracle_unparnmpc_update_solution
        return f(self.h, t, P(arr(q)), P(arr(v)))

    def clear_line_search_filter(self):
        self.lib.oracle_unparnmpc_clear_line_search_filter(self.h)

    def cost_and_violation(self, alpha, q, v):
        out = np.zeros(2)
        self.lib.oracle_unparnmpc_cost_and_violation(self.h, alpha, P(arr(q)), P(arr(v)), P(out))
        return out

    def stage(self, what, t, q, v):
        return self.lib.oracle_unparnmpc_stage(self.h, what, t, P(arr(q)), P(arr(v)))

    def kkt_error(self, t, q, v):
        return self.lib.oracle_unparnmpc_kkt_error(self.h, t, P(arr(q)), P(arr(v)))

    def infeasible_stage(self):
        return self.lib.oracle_unparnmpc_is_current_solution_feasible(self.h)

    def get(self, name):
        out = np.zeros((self.N, self.nv))
        assert self.lib.oracle_unparnmpc_get(self.h, name.encode(), P(out)) == 0, name
        return out

    def step_sizes(self):
        a, b = C.c_double(), C.c_double()
        self.lib.oracle_unparnmpc_get_step_sizes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def matrices(self):
        nv, N = self.nv, self.N
        K, A = np.zeros((N, 5 * nv, 5 * nv)), np.zeros((N, 2 * nv, 2 * nv))
        self.lib.oracle_unparnmpc_get_matrices(self.h, P(K), P(A))
        return K.transpose(0, 2, 1), A.transpose(0, 2, 1)           # col-major -> [row, col]

    def constraint_data(self):
        dimc = 6 * self.nv
        sl, du = np.zeros((self.N, dimc)), np.zeros((self.N, dimc))
        self.lib.oracle_unparnmpc_get_constraint_data(self.h, P(sl), P(du))
        return sl, du

def referee_check(g, o, h, what, tol=1e-10, factor=4.0, window=3):
    """Parity on an ill-conditioned problem, decided by a higher-precision referee.  g, o, h: the same quantity [stages, dim]
    from the GPU, the FP64 oracle and the long double build of the oracle (`hp=True`).  Stage by stage the GPU may be at most
    `factor` times as far from the referee as the FP64 oracle is (largest oracle error within +-`window` stages: rounding
    differences travel along the sweep), plus the 1e-10 bar itself.  Returns (worst GPU error, worst oracle error), relative
    to the largest entry."""
    g, o, h = (np.asarray(x, dtype=np.float64).reshape(len(h), -1) for x in (g, o, h))
    scale = np.maximum(1.0, np.abs(h).max(axis=1))             # per stage, like rel_err
    eg, eo = np.abs(g - h).max(axis=1) / scale, np.abs(o - h).max(axis=1) / scale
    eo_w = np.array([eo[max(0, i - window):i + window + 1].max() for i in range(len(eo))])
    bad = np.nonzero(eg > factor * eo_w + tol)[0]
    assert bad.size == 0, (what, "stage %d: gpu-referee %.2e, oracle-referee %.2e" % (bad[0], eg[bad[0]], eo_w[bad[0]]))
    return eg.max(), eo.max()

def parity(g, o, h, what, tol=1e-10, cap=None):
    """The parity bar of north_star: 1e-10 against the oracle, stage by stage (rel_err).  Where two FP64 evaluation orders cannot
    agree that far the long double referee decides (referee_check: the GPU at most 4x as far from the referee as the FP64 oracle,
    + 1e-10) -- never a loosened tolerance on its own.  h: the referee's value, or a callable that produces it (only evaluated when
    needed).  cap: optional hard limit on the GPU-oracle distance even under the referee rule.  Returns the GPU-oracle distance."""
    e = rel_err(g, o)
    hv = None
    if e >= tol:
        hv = h() if callable(h) else h
        assert hv is not None, (what, "%.2e against the oracle and no referee" % e)
    _parity_table(g, o, hv, what, tol, cap)
    if e >= tol:
        referee_check(g, o, hv, what, tol=tol)
    if cap is not None:
        assert e < cap, (what, e)
    return e


def _parity_table(g, o, hv, what, tol, cap):
    """IDOCP_PARITY_TABLE_DIR=<dir>: every parity() call appends its PER-STAGE error table to <dir>/parity_tables.txt -- the GPU-oracle distance
    of every stage (relative to the stage's largest entry, like rel_err) and, where the referee was consulted, both distances from it -- so that a
    regression INSIDE a cap or inside the referee rule's slack is visible in the committed copy (profiles/rNN_parity_tables.txt)."""
    d = os.environ.get("IDOCP_PARITY_TABLE_DIR")
    if not d:
        return
    ga, oa = np.asarray(g, dtype=np.float64), np.asarray(o, dtype=np.float64)
    if ga.ndim < 2 or ga.shape != oa.shape:
        return
    n = ga.shape[0]
    ga, oa = ga.reshape(n, -1), oa.reshape(n, -1)
    scale = np.maximum(1.0, np.abs(oa).max(axis=1))
    ego = np.abs(ga - oa).max(axis=1) / scale
    # the same distance ENTRY BY ENTRY (relative to the entry itself) over the entries that are at least 1e-3 of their stage's scale (the
    # largest entry, never less than 1 -- the scale rel_err uses): rel_err holds the smaller entries of a large stage to an absolute bar, this
    # column shows what that hides among the entries that carry the stage (a diagnostic, not a bar)
    big = np.abs(oa) >= 1e-3 * scale[:, None]
    cw = np.where(big, np.abs(ga - oa) / np.maximum(np.abs(oa), 1e-3), 0.0)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_tables.txt"), "a") as f:
        f.write("## [%s] %s   tol %.0e  cap %s  worst %.2e at stage %d  (%d stages, %d above tol)  entry-wise over the entries >= 1e-3 of the stage's scale: %.2e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], what, tol, "%.0e" % cap if cap else "-", ego.max(), int(ego.argmax()), n, int((ego >= tol).sum()), float(cw.max()) if cw.size else 0.0))
        if hv is not None:
            ha = np.asarray(hv, dtype=np.float64).reshape(n, -1)
            sh = np.maximum(1.0, np.abs(ha).max(axis=1))
            egh, eoh = np.abs(ga - ha).max(axis=1) / sh, np.abs(oa - ha).max(axis=1) / sh
            f.write("# stage  gpu-oracle  gpu-referee  oracle-referee   (only stages above tol / 10)\n")
            for i in range(n):
                if ego[i] >= tol / 10:
                    f.write("%6d  %.2e  %.2e  %.2e\n" % (i, ego[i], egh[i], eoh[i]))
        else:
            worst = np.argsort(ego)[::-1][:5]
            f.write("# five worst stages (gpu-oracle): " + "  ".join("%d: %.2e" % (int(i), ego[i]) for i in worst) + "\n")

def pairwise_check(a, b, o, h, what, tol=1e-10, factor=4.0, window=3):
    """Two GPU evaluations a, b of the same quantity (e.g. two instantiations of a kernel) against EACH OTHER, stage by stage, at the
    bar the referee rule implies for the pair: on the stages where the FP64 oracle o sits on its long double build h (to 1e-11, over
    +-`window` stages) the two must agree to 2 * tol -- no cap hides a regression there --, elsewhere to 2 * (factor * oracle error + tol)."""
    a, b, o, h = (np.asarray(x, dtype=np.float64).reshape(len(h), -1) for x in (a, b, o, h))
    scale = np.maximum(1.0, np.abs(h).max(axis=1))
    eo = np.abs(o - h).max(axis=1) / scale
    eo_w = np.array([eo[max(0, i - window):i + window + 1].max() for i in range(len(eo))])
    d = np.abs(a - b).max(axis=1) / scale
    bound = 2.0 * (np.where(eo_w < 1e-11, 0.0, factor * eo_w) + tol)
    bad = np.nonzero(d > bound)[0]
    assert bad.size == 0, (what, "stage %d: the two differ by %.2e, bound %.2e (oracle-referee %.2e)" % (bad[0], d[bad[0]], bound[bad[0]], eo_w[bad[0]]))
    return float(d.max()), int((eo_w < 1e-11).sum())

def rel_err(a, b):
    """Largest deviation of a from b, STAGE BY STAGE: arrays with a leading stage (or instance) axis are compared entry-wise and
    every stage's error is taken relative to the largest entry of that very stage of b (never less than 1: entries below one are
    held to the absolute bar).  Along a trotting chain |P| and the direction vary by orders of magnitude from stage to stage; a
    norm over the whole horizon would let the large stages hide an error in the small ones."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.ndim < 2 or a.shape != b.shape:
        return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    n = a.shape[0]
    err = np.abs(a - b).reshape(n, -1).max(axis=1)
    scale = np.maximum(1.0, np.abs(b).reshape(n, -1).max(axis=1))
    return float((err / scale).max())


# ------------------------------------------------------------------ contact path

def _setup_oracle_ocp(lib):
    if getattr(lib, "_ocp_ready", False):
        return
    vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
    PM = C.POINTER(capi.Model)
    lib.oracle_ocp_create.argtypes = [PM, C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci]
    lib.oracle_ocp_create.restype = vp
    lib.oracle_ocp_destroy.argtypes = [vp]
    lib.oracle_ocp_set_contact_status.argtypes = [vp, C.POINTER(ci), dp]
    lib.oracle_ocp_set_solution.argtypes = [vp, cs, dp]
    lib.oracle_ocp_init_constraints.argtypes = [vp, cd]
    lib.oracle_ocp_update_solution.argtypes = [vp, cd, dp, dp]
    lib.oracle_ocp_stage.argtypes = [vp, ci, cd, dp, dp]
    lib.oracle_ocp_compute_kkt_residual.argtypes = [vp, cd, dp, dp]
    lib.oracle_ocp_kkt_error.argtypes = [vp]
    lib.oracle_ocp_kkt_error.restype = cd
    lib.oracle_ocp_is_current_solution_feasible.argtypes = [vp]
    lib.oracle_ocp_q_ref.argtypes = [vp, cd, dp]
    lib.oracle_ocp_get.argtypes = [vp, cs, ci, dp]
    lib.oracle_ocp_get_step_sizes.argtypes = [vp, dp, dp]
    lib.oracle_ocp_get_riccati.argtypes = [vp, dp, dp, dp, dp]
    lib.oracle_ocp_dimc.argtypes = [vp]
    lib.oracle_ocp_get_constraint_data.argtypes = [vp, dp, dp]
    lib.oracle_ocp_get_lqr_stage.argtypes = [vp, ci] + [dp] * 8
    lib.oracle_ocp_bench.argtypes = [vp, cd, dp, dp, ci, dp]
    lib.oracle_ocp_bench.restype = cd
    lib.oracle_ocp_set_num_threads.argtypes = [vp, ci]
    lib.oracle_ocp_create_hybrid.argtypes = [PM, C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci, ci]
    lib.oracle_ocp_create_hybrid.restype = vp
    lib.oracle_ocp_push_back_contact_status.argtypes = [vp, C.POINTER(ci), dp, cd]
    lib.oracle_ocp_set_contact_points.argtypes = [vp, ci, dp]
    ip = C.POINTER(ci)
    lib.oracle_ocp_chain.argtypes = [vp, cd, ip, ip, ip, dp, dp, ip, ip]
    lib.oracle_ocp_get_chain.argtypes = [vp, cs, ci, dp]
    lib.oracle_ocp_get_riccati_chain.argtypes = [vp, dp, dp, dp, dp]
    lib._ocp_ready = True

class OracleOCP:
    def __init__(self, model, cost, cons, T, N, max_num_impulse=0, hp=False):
        self.lib = oracle(hp)
        _setup_oracle_ocp(self.lib)
        self.N, self.nv, self.nu, self.nq = N, model.nv, model.nu, model.nq
        self.max_events = max_num_impulse
        if max_num_impulse > 0:
            self.h = self.lib.oracle_ocp_create_hybrid(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse)
        else:
            self.h = self.lib.oracle_ocp_create(C.byref(model), C.byref(cost), C.byref(cons), T, N)
        assert self.h

    # ---- contact sequences with discrete events (OCPSolver::pushBackContactStatus, ocp_solver.cpp:174-184)
    def push_back_contact_status(self, active, points, switching_time):
        a = (C.c_int * 4)(*[int(x) for x in active])
        assert self.lib.oracle_ocp_push_back_contact_status(self.h, a, P(arr(points)), switching_time) == 0

    def set_contact_points(self, phase, points):
        assert self.lib.oracle_ocp_set_contact_points(self.h, phase, P(arr(points))) == 0

    # OCPSolver::popBackContactStatus / popFrontContactStatus (ocp_solver.cpp:187-194)
    def pop_back_contact_status(self):
        self.lib.oracle_ocp_pop_back_contact_status.argtypes = [C.c_void_p]
        assert self.lib.oracle_ocp_pop_back_contact_status(self.h) == 0

    def pop_front_contact_status(self):
        self.lib.oracle_ocp_pop_front_contact_status.argtypes = [C.c_void_p]
        assert self.lib.oracle_ocp_pop_front_contact_status(self.h) == 0

    def chain(self, t):
        """Stages in time order after OCPDiscretizer::discretizeOCP(t): list of dicts."""
        cap = self.N + 1 + 3 * max(self.max_events, 1)
        IA = lambda: (C.c_int * cap)()
        kind, index, slot, sw, dimf = IA(), IA(), IA(), IA(), IA()
        tt, dt = np.zeros(cap), np.zeros(cap)
        M = self.lib.oracle_ocp_chain(self.h, t, kind, index, slot, P(tt), P(dt), sw, dimf)
        return [dict(kind=NODE_KINDS[kind[p]], index=index[p], slot=slot[p], t=tt[p], dt=dt[p], sw_event=sw[p], dimf=dimf[p])
                for p in range(M)]

    def set_task_refs(self, times, refs):
        """TimeVarying task-space cost: the reference poses refs[M][12] tabulated at the stage times (looked up by time)."""
        times, refs = np.ascontiguousarray(times, dtype=np.float64), np.ascontiguousarray(refs, dtype=np.float64)
        self.lib.oracle_ocp_set_task_refs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        assert self.lib.oracle_ocp_set_task_refs(self.h, len(times), P(times), P(refs)) == 0

    def get_chain(self, name, M):
        dim = OCP_SOL_FIELDS.get(name) or OCP_DIR_FIELDS.get(name) or OCP_CHAIN_EXTRA[name]
        out = np.zeros((M, dim))
        assert self.lib.oracle_ocp_get_chain(self.h, name.encode(), dim, P(out)) == 0
        return out

    def riccati_chain(self, M):
        nv, nu = self.nv, self.nu
        Pm, s = np.zeros((M, 2 * nv, 2 * nv)), np.zeros((M, 2 * nv))
        K, k = np.zeros((M - 1, 2 * nv, nu)), np.zeros((M - 1, nu))
        self.lib.oracle_ocp_get_riccati_chain(self.h, P(Pm), P(s), P(K), P(k))
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_ocp_destroy(self.h)
            self.h = None

    def set_contact_status(self, active, points):
        a = (C.c_int * 4)(*[int(x) for x in active])
        assert self.lib.oracle_ocp_set_contact_status(self.h, a, P(arr(points))) == 0

    def set_solution(self, name, value):
        assert self.lib.oracle_ocp_set_solution(self.h, name.encode(), P(arr(value))) == 0

    def init_constraints(self, t=0.0):
        self.lib.oracle_ocp_init_constraints(self.h, t)

    def update(self, t, q, v):
        return self.lib.oracle_ocp_update_solution(self.h, t, P(arr(q)), P(arr(v)))

    def stage(self, what, t, q, v):
        return self.lib.oracle_ocp_stage(self.h, what, t, P(arr(q)), P(arr(v)))

    def infeasible_stage(self):
        """chain position of the first stage violating an inequality constraint, -1 if the iterate is feasible"""
        return self.lib.oracle_ocp_is_current_solution_feasible(self.h)

    def kkt_error(self, t, q, v):
        self.lib.oracle_ocp_compute_kkt_residual(self.h, t, P(arr(q)), P(arr(v)))
        return self.lib.oracle_ocp_kkt_error(self.h)

    def _dim(self, name):
        # ANYmal: the table; any other robot (the fixed-base arm of tests/test_oracle_fixed_base.py): from the model's dimensions
        if self.nv == 18:
            return OCP_SOL_FIELDS.get(name) or OCP_DIR_FIELDS[name]
        return {"q": self.nq, "u": self.nu, "du": self.nu}.get(name, self.nv)

    def get(self, name):
        dim = self._dim(name)
        out = np.zeros((self.N + 1, dim))
        assert self.lib.oracle_ocp_get(self.h, name.encode(), dim, P(out)) == 0
        return out[:self.N] if name in OCP_STAGE_ONLY else out

    def step_sizes(self):
        a, b = C.c_double(), C.c_double()
        self.lib.oracle_ocp_get_step_sizes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def riccati(self):
        nv, nu, N = self.nv, self.nu, self.N
        Pm, s = np.zeros((N + 1, 2 * nv, 2 * nv)), np.zeros((N + 1, 2 * nv))
        K, k = np.zeros((N, 2 * nv, nu)), np.zeros((N, nu))
        self.lib.oracle_ocp_get_riccati(self.h, P(Pm), P(s), P(K), P(k))
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k

    def constraint_data(self):
        dimc = self.lib.oracle_ocp_dimc(self.h)
        sl, du = np.zeros((self.N, dimc)), np.zeros((self.N, dimc))
        self.lib.oracle_ocp_get_constraint_data(self.h, P(sl), P(du))
        return sl, du

    def lqr_stage(self, i):
        nv, nu = self.nv, self.nu
        nx = 2 * nv
        Qxx, Qxu, Quu, A, B = np.zeros((nx, nx)), np.zeros((nu, nx)), np.zeros((nu, nu)), np.zeros((nx, nx)), np.zeros((nu, nx))
        lx, lu, Fx = np.zeros(nx), np.zeros(nu), np.zeros(nx)
        self.lib.oracle_ocp_get_lqr_stage(self.h, i, P(Qxx), P(Qxu), P(Quu), P(A), P(B), P(lx), P(lu), P(Fx))
        return Qxx.T, Qxu.T, Quu.T, A.T, B.T, lx, lu, Fx

class OracleParNMPC:
    """ParNMPCSolver of the oracle: examples/anymal/parnmpc_benchmark.cpp call order; with max_num_impulse > 0 also horizons
    with discrete events (examples/anymal/anymal_trotting_parnmpc.cpp)."""
    KINDS = ("stage", "impulse", "aux", "lift", "terminal")

    def __init__(self, model, cost, cons, T, N, max_num_impulse=0, hp=False):
        self.lib = lib = oracle(hp)
        vp, ci, cd, cs = C.c_void_p, C.c_int, C.c_double, C.c_char_p
        if not getattr(lib, "_parnmpc_ready", False):
            lib.oracle_parnmpc_create.argtypes = [C.POINTER(capi.Model), C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci]
            lib.oracle_parnmpc_create.restype = vp
            lib.oracle_parnmpc_destroy.argtypes = [vp]
            lib.oracle_parnmpc_set_contact_status.argtypes = [vp, C.POINTER(ci), dp]
            lib.oracle_parnmpc_set_solution.argtypes = [vp, cs, dp]
            lib.oracle_parnmpc_init.argtypes = [vp, cd]
            lib.oracle_parnmpc_update_solution.argtypes = [vp, cd, dp, dp]
            lib.oracle_parnmpc_update_solution_ls.argtypes = [vp, cd, dp, dp]
            lib.oracle_parnmpc_compute_direction.argtypes = [vp, cd, dp, dp]
            lib.oracle_parnmpc_cost_and_violation.argtypes = [vp, cd, dp, dp, dp]
            lib.oracle_parnmpc_clear_line_search_filter.argtypes = [vp]
            lib.oracle_parnmpc_kkt_error.argtypes = [vp, cd, dp, dp]
            lib.oracle_parnmpc_kkt_error.restype = cd
            lib.oracle_parnmpc_is_current_solution_feasible.argtypes = [vp]
            lib.oracle_parnmpc_get.argtypes = [vp, cs, ci, dp]
            lib.oracle_parnmpc_get_step_sizes.argtypes = [vp, dp, dp]
            lib.oracle_parnmpc_set_shard.argtypes = [vp, ci, ci, ci]
            lib.oracle_parnmpc_phase.argtypes = [vp, ci, cd, dp, dp]
            lib.oracle_parnmpc_halo_size.argtypes = [vp, ci]
            lib.oracle_parnmpc_export.argtypes = [vp, ci, dp]
            lib.oracle_parnmpc_import.argtypes = [vp, ci, dp]
            lib.oracle_parnmpc_set_step_sizes.argtypes = [vp, cd, cd]
            lib.oracle_parnmpc_kkt_error_squared.argtypes = [vp, cd, dp, dp]
            lib.oracle_parnmpc_kkt_error_squared.restype = cd
            lib.oracle_parnmpc_init_constraints_only.argtypes = [vp, cd]
            lib.oracle_parnmpc_init_aux_only.argtypes = [vp, cd]
            lib._parnmpc_ready = True
            lib.oracle_parnmpc_create_hybrid.argtypes = [C.POINTER(capi.Model), C.POINTER(capi.Cost), C.POINTER(capi.Constraints), cd, ci, ci]
            lib.oracle_parnmpc_create_hybrid.restype = vp
            lib.oracle_parnmpc_push_back_contact_status.argtypes = [vp, C.POINTER(ci), dp, cd]
            ip = C.POINTER(ci)
            lib.oracle_parnmpc_chain.argtypes = [vp, cd, ci, ip, ip, ip, dp, dp, ip, ip]
            lib.oracle_parnmpc_get_chain.argtypes = [vp, cs, ci, dp]
        self.N, self.nv, self.max_num_impulse = N, model.nv, max_num_impulse
        self.nq, self.nu = model.nq, model.nu
        if max_num_impulse > 0:
            self.h = lib.oracle_parnmpc_create_hybrid(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse)
        else:
            self.h = lib.oracle_parnmpc_create(C.byref(model), C.byref(cost), C.byref(cons), T, N)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_parnmpc_destroy(self.h)
            self.h = None

    def set_contact_status(self, active, points):
        a = (C.c_int * 4)(*[int(x) for x in active])
        assert self.lib.oracle_parnmpc_set_contact_status(self.h, a, P(arr(points))) == 0

    def set_solution(self, name, value):
        assert self.lib.oracle_parnmpc_set_solution(self.h, name.encode(), P(arr(value))) == 0

    def push_back_contact_status(self, active, points, switching_time):
        a = (C.c_int * 4)(*[int(x) for x in active])
        assert self.lib.oracle_parnmpc_push_back_contact_status(self.h, a, P(arr(points)), switching_time) == 0

    def pop_back_contact_status(self):
        self.lib.oracle_parnmpc_pop_back_contact_status.argtypes = [C.c_void_p]
        assert self.lib.oracle_parnmpc_pop_back_contact_status(self.h) == 0

    def pop_front_contact_status(self):
        self.lib.oracle_parnmpc_pop_front_contact_status.argtypes = [C.c_void_p]
        assert self.lib.oracle_parnmpc_pop_front_contact_status(self.h) == 0

    def chain(self, t=0.0):
        cap = self.N + 3 * self.max_num_impulse + 1
        I = lambda: (C.c_int * cap)()
        kind, index, slot, dimf, level = I(), I(), I(), I(), I()
        tt, dt = np.zeros(cap), np.zeros(cap)
        M = self.lib.oracle_parnmpc_chain(self.h, t, cap, kind, index, slot, P(tt), P(dt), dimf, level)
        assert M >= 0, "the oracle rejected the discretisation"
        return [dict(kind=self.KINDS[kind[p]], index=index[p], slot=slot[p], t=tt[p], dt=dt[p], dimf=dimf[p], level=level[p]) for p in range(M)]

    def set_task_refs(self, times, refs):
        times, refs = np.ascontiguousarray(times, dtype=np.float64), np.ascontiguousarray(refs, dtype=np.float64)
        self.lib.oracle_parnmpc_set_task_refs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        assert self.lib.oracle_parnmpc_set_task_refs(self.h, len(times), P(times), P(refs)) == 0

    def get_chain(self, name, M):
        base = name[4:] if name.startswith("new_") else name      # "new_" + field: the coarse / corrected iterate of the backward correction
        dim = OCP_SOL_FIELDS.get(base) or OCP_DIR_FIELDS.get(base) or {"xi": 12, "dxi": 12}[base]
        out = np.zeros((M, dim))
        assert self.lib.oracle_parnmpc_get_chain(self.h, name.encode(), dim, P(out)) == 0
        return out

    def set_stage_values(self, name, values):
        self.lib.oracle_parnmpc_set_stage.argtypes = [C.c_void_p, C.c_int, C.c_char_p, dp]
        for i, v in enumerate(np.asarray(values)):
            assert self.lib.oracle_parnmpc_set_stage(self.h, i, name.encode(), P(arr(v))) == 0

    def set_aux_mats(self, mats):
        self.lib.oracle_parnmpc_set_aux_mat.argtypes = [C.c_void_p, C.c_int, dp]
        for i, mth in enumerate(np.asarray(mats)):
            assert self.lib.oracle_parnmpc_set_aux_mat(self.h, i, P(arr(np.asarray(mth).T))) == 0

    def set_chain_values(self, name, values):
        """warm start along the chain of the current discretisation: entry p goes to the slot of chain position p"""
        self.lib.oracle_parnmpc_set_stage.argtypes = [C.c_void_p, C.c_int, C.c_char_p, dp]
        ch = self.chain(0.0)
        for c, v in zip(ch, np.asarray(values)):
            assert self.lib.oracle_parnmpc_set_stage(self.h, c["slot"], name.encode(), P(arr(v))) == 0

    def set_chain_aux_mats(self, mats):
        self.lib.oracle_parnmpc_set_aux_mat.argtypes = [C.c_void_p, C.c_int, dp]
        ch = self.chain(0.0)
        for c, mth in zip(ch, np.asarray(mats)):
            assert self.lib.oracle_parnmpc_set_aux_mat(self.h, c["slot"], P(arr(np.asarray(mth).T))) == 0

    def init(self, t=0.0):                      # initBackwardCorrection(t) + initConstraints(t)
        self.lib.oracle_parnmpc_init(self.h, t)

    def init_constraints(self, t=0.0):          # initConstraints(t) alone (after a warm start of the iterate)
        self.lib.oracle_parnmpc_init_constraints_only(self.h, t)

    def update(self, t, q, v):
        return self.lib.oracle_parnmpc_update_solution(self.h, t, P(arr(q)), P(arr(v)))

    def infeasible_stage(self):
        return self.lib.oracle_parnmpc_is_current_solution_feasible(self.h)

    def kkt_error(self, t, q, v):
        return self.lib.oracle_parnmpc_kkt_error(self.h, t, P(arr(q)), P(arr(v)))

    def get(self, name):
        dim = (OCP_SOL_FIELDS.get(name) or OCP_DIR_FIELDS[name]) if self.nv == 18 else {"q": self.nq, "u": self.nu, "du": self.nu}.get(name, self.nv)
        out = np.zeros((self.N, dim))
        assert self.lib.oracle_parnmpc_get(self.h, name.encode(), dim, P(out)) == 0
        return out

    def step_sizes(self):
        a, b = C.c_double(), C.c_double()
        self.lib.oracle_parnmpc_get_step_sizes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

class OracleParNMPCShard:
    """Shard backend of tests/parnmpc_dist.py ShardedParNMPC on top of the oracle (one instance, CPU tensors)."""
    PHASES = {"linearize": 0, "bwd_serial": 1, "bwd_parallel": 2, "fwd_serial": 3, "fwd_parallel": 4, "integrate": 5}

    def __init__(self, model, cost, cons, T, N, rank, world, q0, v0, max_num_impulse=0):
        assert N % world == 0
        self.Nl = N // world
        if max_num_impulse > 0:
            # a horizon with discrete events: every shard discretises the whole horizon and keeps its slice of the chain
            self.o = OracleParNMPC(model, cost, cons, T, N, max_num_impulse=max_num_impulse)
            self.o.lib.oracle_parnmpc_set_chain_slice.argtypes = [C.c_void_p, C.c_int, C.c_int]
            self.o.lib.oracle_parnmpc_set_chain_slice(self.o.h, rank * self.Nl, (rank + 1) * self.Nl)
        else:
            self.o = OracleParNMPC(model, cost, cons, T / world, self.Nl)
            self.o.lib.oracle_parnmpc_set_shard(self.o.h, rank * self.Nl, 1 if rank == world - 1 else 0, 1 if rank > 0 else 0)
        self.batch = 1
        self.q_prev, self.v_prev = arr(q0).copy(), arr(v0).copy()      # rank 0: the measured state; else the imported halo
        self.nq = model.nq

    def halo_size(self, kind):
        return self.o.lib.oracle_parnmpc_halo_size(self.o.h, kind)

    def export(self, kind):
        import torch
        out = np.zeros(self.halo_size(kind))
        self.o.lib.oracle_parnmpc_export(self.o.h, kind, P(out))
        return torch.from_numpy(out).reshape(1, -1)

    def import_(self, kind, tensor):
        a = np.ascontiguousarray(tensor.numpy().reshape(-1))
        self.o.lib.oracle_parnmpc_import(self.o.h, kind, P(a))
        if kind == 0:
            self.q_prev, self.v_prev = a[:self.nq].copy(), a[self.nq:].copy()

    def phase(self, name, t):
        if name == "init_aux":
            self.o.lib.oracle_parnmpc_init_aux_only(self.o.h, t)
            return
        assert self.o.lib.oracle_parnmpc_phase(self.o.h, self.PHASES[name], t, P(self.q_prev), P(self.v_prev)) == 0

    def local_steps(self):
        import torch
        a, b = self.o.step_sizes()
        return torch.tensor([[a, b]], dtype=torch.float64)

    def set_steps(self, tensor):
        self.o.lib.oracle_parnmpc_set_step_sizes(self.o.h, float(tensor[0, 0]), float(tensor[0, 1]))

    def err2(self, t):
        import torch
        return torch.tensor([self.o.lib.oracle_parnmpc_kkt_error_squared(self.o.h, t, P(self.q_prev), P(self.v_prev))], dtype=torch.float64)

class OracleUnParNMPCShard:
    """Shard backend of tests/parnmpc_dist.py ShardedParNMPC on top of the fixed-base oracle (one instance, CPU tensors):
    every rank holds a whole-horizon oracle and works on its slice of the stages; the halos fill the neighbours' stages."""
    PHASES = {"linearize": 0, "bwd_serial": 1, "bwd_parallel": 2, "fwd_serial": 3, "fwd_parallel": 4, "integrate": 5}

    def __init__(self, model, cost, cons, T, N, rank, world, q0, v0):
        assert N % world == 0
        self.Nl, self.rank = N // world, rank
        self.o = OracleUnParNMPC(model, cost, cons, T, N)
        self.o.lib.oracle_unparnmpc_set_slice(self.o.h, rank * self.Nl, (rank + 1) * self.Nl)
        self.batch, self.nv = 1, model.nv
        self.q0, self.v0 = arr(q0).copy(), arr(v0).copy()          # the measured state (used by rank 0 only)

    def halo_size(self, kind):
        return {0: 2 * self.nv, 1: 2 * self.nv, 2: 4 * self.nv * self.nv, 3: 2 * self.nv, 4: 2 * self.nv, 5: 1}[kind]

    def export(self, kind):
        import torch
        out = np.zeros(self.halo_size(kind))
        if kind != 5:
            assert self.o.lib.oracle_unparnmpc_export(self.o.h, kind, P(out)) == 0
        return torch.from_numpy(out).reshape(1, -1)

    def import_(self, kind, tensor):
        if kind == 5:
            return
        a = np.ascontiguousarray(tensor.numpy().reshape(-1))
        assert self.o.lib.oracle_unparnmpc_import(self.o.h, kind, P(a)) == 0

    def phase(self, name, t):
        if name == "init_aux":
            self.o.init(t)
            return
        assert self.o.stage(self.PHASES[name], t, self.q0, self.v0) == 0

    def local_steps(self):
        import torch
        a, b = self.o.step_sizes()
        return torch.tensor([[a, b]], dtype=torch.float64)

    def set_steps(self, tensor):
        self.o.lib.oracle_unparnmpc_set_step_sizes(self.o.h, float(tensor[0, 0]), float(tensor[0, 1]))

    def err2(self, t):
        import torch
        return torch.tensor([self.o.lib.oracle_unparnmpc_kkt_error_squared(self.o.h, t, P(self.q0), P(self.v0))], dtype=torch.float64)
