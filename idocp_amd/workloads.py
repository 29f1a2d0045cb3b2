"""Workloads of the BASELINE configs and the Python mirror of the solver classes over the C ABI.

Everything bench.py, __graft_entry__.smoke() and the tests share that is NOT test infrastructure: the robot models (URDF fixtures of
the reference's own tests under tests/golden/urdf), the cost / constraint definitions of the reference's examples transcribed as
data (examples/iiwa14/*.cpp, examples/anymal/*.cpp), the contact sequences of the trotting and running gaits, and thin
ctypes wrappers (HipUnOCP, HipOCP, HipParNMPC, ...) with the reference's method names.  No arithmetic lives here; the oracle
wrappers and the parity checkers are in tests/helpers.py.
"""
import ctypes as C
import json
import os

import numpy as np

from . import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")

IIWA_URDF = os.path.join(GOLDEN, "urdf", "iiwa14.urdf")

ANYMAL_URDF = os.path.join(GOLDEN, "urdf", "anymal.urdf")

ANYMAL_CONTACT_FRAMES = (14, 24, 34, 44)

dp = capi.c_double_p

def arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))

def P(a):
    return a.ctypes.data_as(dp)

def iiwa14_model():
    return capi.model_from_urdf(IIWA_URDF)

def unocp_problem(model):
    """Workload of examples/iiwa14/unocp_benchmark.cpp:20-43."""
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", np.full(nv, -5.0)).set("v_ref", np.full(nv, -9.0))
    cost.set("q_weight", np.full(nv, 10.0)).set("qf_weight", np.full(nv, 10.0))
    cost.set("v_weight", np.full(nv, 0.1)).set("vf_weight", np.full(nv, 0.1))
    cost.set("a_weight", np.full(nv, 0.01)).set("u_weight", np.zeros(nv))
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    for i in range(model.nu):           # robot.setJointEffortLimit(Constant(200))
        model.u_max[i] = 200.0
    return cost, cons

SOL_FIELDS = ("q", "v", "a", "u", "lmd", "gmm", "beta")

DIR_FIELDS = tuple("d" + f for f in SOL_FIELDS)

def stages_of(name, N):
    return N if name in ("a", "u", "beta", "da", "du", "dbeta") else N + 1

class HipUnOCP:
    """Product path through the C ABI (idocp_unocp_*)."""

    def __init__(self, model, cost, cons, T, N, batch=1, device=0):
        self.lib = capi.lib()
        self.N, self.nv, self.batch = N, model.nv, batch
        h = C.c_void_p()
        capi.check(self.lib.idocp_unocp_create(C.byref(model), C.byref(cost), C.byref(cons), T, N, batch, device, C.byref(h)),
                   "idocp_unocp_create")
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.idocp_unocp_destroy(self.h)
            self.h = None

    def set_solution(self, name, value):
        capi.check(self.lib.idocp_unocp_set_solution(self.h, name.encode(), P(arr(value))), "set_solution")

    def set_solution_batch(self, name, values):
        capi.check(self.lib.idocp_unocp_set_solution_batch(self.h, name.encode(), P(arr(values))), "set_solution_batch")

    def update(self, t, q, v, line_search=False):
        q = np.broadcast_to(arr(q), (self.batch, self.nv)) if np.ndim(q) == 1 else q
        v = np.broadcast_to(arr(v), (self.batch, self.nv)) if np.ndim(v) == 1 else v
        return self.lib.idocp_unocp_update_solution(self.h, t, P(arr(q)), P(arr(v)), 1 if line_search else 0)

    def set_task_refs(self, refs):
        """reference poses of a TimeVaryingTaskSpace*Cost at the stage times, [N + 1][12] (rotation row-major, position)"""
        refs = arr(refs)
        assert refs.shape == (self.N + 1, 12)
        capi.check(self.lib.idocp_unocp_set_task_refs(self.h, P(refs)), "set_task_refs")

    def launch(self, kernel_id, q, v):
        """one kernel of updateSolution (0 linearize, 1 / 2 Riccati backward / forward, 3 expand, 4 reduce steps, 5 integrate)"""
        if getattr(self, "_dq", None) is None:
            self._dq, self._dv = C.c_void_p(), C.c_void_p()
            capi.check(self.lib.idocp_device_alloc(C.byref(self._dq), 8 * self.batch * self.nv), "alloc")
            capi.check(self.lib.idocp_device_alloc(C.byref(self._dv), 8 * self.batch * self.nv), "alloc")
        qq = arr(np.broadcast_to(arr(q), (self.batch, self.nv)) if np.ndim(q) == 1 else q)
        vv = arr(np.broadcast_to(arr(v), (self.batch, self.nv)) if np.ndim(v) == 1 else v)
        capi.check(self.lib.idocp_device_upload(self._dq, qq.ctypes.data, qq.nbytes), "upload")
        capi.check(self.lib.idocp_device_upload(self._dv, vv.ctypes.data, vv.nbytes), "upload")
        capi.check(self.lib.idocp_unocp_launch_kernel(self.h, kernel_id, self._dq, self._dv), "launch_kernel")
        capi.check(self.lib.idocp_unocp_synchronize(self.h), "synchronize")

    def clear_line_search_filter(self):
        capi.check(self.lib.idocp_unocp_clear_line_search_filter(self.h), "clear_line_search_filter")

    def cost_and_violation(self, alpha):
        """UnLineSearch::computeCostAndViolation at s + alpha d (scalar or per instance) -> (cost[batch], violation[batch])"""
        a = arr(np.broadcast_to(arr(alpha), (self.batch,)))
        c, v = np.zeros(self.batch), np.zeros(self.batch)
        capi.check(self.lib.idocp_unocp_line_search_eval(self.h, P(a), P(c), P(v)), "line_search_eval")
        return c, v

    def infeasible_stage(self):
        ok, where = np.zeros(self.batch, dtype=np.int32), np.zeros(self.batch, dtype=np.int32)
        capi.check(self.lib.idocp_unocp_is_current_solution_feasible(self.h, ok.ctypes.data_as(capi.c_int_p),
                                                                     where.ctypes.data_as(capi.c_int_p)), "is_feasible")
        assert np.all((where < 0) == (ok == 1))
        return where

    def kkt_error(self, t, q, v):
        q = np.broadcast_to(arr(q), (self.batch, self.nv)) if np.ndim(q) == 1 else q
        v = np.broadcast_to(arr(v), (self.batch, self.nv)) if np.ndim(v) == 1 else v
        capi.check(self.lib.idocp_unocp_compute_kkt_residual(self.h, t, P(arr(q)), P(arr(v))), "compute_kkt_residual")
        out = np.zeros(self.batch)
        capi.check(self.lib.idocp_unocp_kkt_error(self.h, P(out)), "kkt_error")
        return out

    def solution(self, name, instance=0):
        out = np.zeros((self.N + 1, self.nv))
        capi.check(self.lib.idocp_unocp_get_solution(self.h, name.encode(), instance, P(out)), "get_solution")
        return out[:stages_of(name, self.N)]

    def direction(self, name, instance=0):
        out = np.zeros((self.N + 1, self.nv))
        capi.check(self.lib.idocp_unocp_get_direction(self.h, name.encode(), instance, P(out)), "get_direction")
        return out[:stages_of(name, self.N)]

    def step_sizes(self):
        a, b = np.zeros(self.batch), np.zeros(self.batch)
        capi.check(self.lib.idocp_unocp_get_step_sizes(self.h, P(a), P(b)), "get_step_sizes")
        return a, b

    def riccati(self, instance=0):
        nv, N = self.nv, self.N
        Pm, s = np.zeros((N + 1, 2 * nv, 2 * nv)), np.zeros((N + 1, 2 * nv))
        K, k = np.zeros((N, 2 * nv, nv)), np.zeros((N, nv))
        capi.check(self.lib.idocp_unocp_get_riccati(self.h, instance, P(Pm), P(s), P(K), P(k)), "get_riccati")
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k

    def constraint_data(self, instance=0):
        dimc = self.lib.idocp_unocp_dimc(self.h)
        sl, du = np.zeros((self.N, dimc)), np.zeros((self.N, dimc))
        capi.check(self.lib.idocp_unocp_get_constraint_data(self.h, instance, P(sl), P(du)), "get_constraint_data")
        return sl, du

class HipUnParNMPC(HipUnOCP):
    """Product path of UnParNMPCSolver through the C ABI (idocp_unparnmpc_* on the shared handle type); fields are [N][nv]."""

    def __init__(self, model, cost, cons, T, N, batch=1, device=0):
        self.lib = capi.lib()
        self.N, self.nv, self.batch = N, model.nv, batch
        h = C.c_void_p()
        capi.check(self.lib.idocp_unparnmpc_create(C.byref(model), C.byref(cost), C.byref(cons), T, N, batch, device, C.byref(h)),
                   "idocp_unparnmpc_create")
        self.h = h
        self._dq = self._dv = None

    def init(self, t=0.0):
        capi.check(self.lib.idocp_unocp_init_constraints(self.h), "init_constraints")
        capi.check(self.lib.idocp_unparnmpc_init_backward_correction(self.h, t), "init_backward_correction")

    def _bc(self, x):
        return arr(np.broadcast_to(arr(x), (self.batch, self.nv)) if np.ndim(x) == 1 else x)

    def update(self, t, q, v, line_search=False):
        return self.lib.idocp_unparnmpc_update_solution(self.h, t, P(self._bc(q)), P(self._bc(v)), 1 if line_search else 0)

    def phase(self, phase, q, v):
        """one phase of updateSolution (0 linearize ... 6 integrate) with (q, v) uploaded once"""
        if self._dq is None:
            self._dq, self._dv = C.c_void_p(), C.c_void_p()
            capi.check(self.lib.idocp_device_alloc(C.byref(self._dq), 8 * self.batch * self.nv), "alloc")
            capi.check(self.lib.idocp_device_alloc(C.byref(self._dv), 8 * self.batch * self.nv), "alloc")
        qq, vv = self._bc(q), self._bc(v)
        capi.check(self.lib.idocp_device_upload(self._dq, qq.ctypes.data, qq.nbytes), "upload")
        capi.check(self.lib.idocp_device_upload(self._dv, vv.ctypes.data, vv.nbytes), "upload")
        capi.check(self.lib.idocp_unparnmpc_launch_phase(self.h, phase, self._dq, self._dv), "launch_phase")
        capi.check(self.lib.idocp_unocp_synchronize(self.h), "synchronize")

    def kkt_error(self, t, q, v):
        capi.check(self.lib.idocp_unparnmpc_compute_kkt_residual(self.h, t, P(self._bc(q)), P(self._bc(v))), "compute_kkt_residual")
        out = np.zeros(self.batch)
        capi.check(self.lib.idocp_unocp_kkt_error(self.h, P(out)), "kkt_error")
        return out

    def get(self, name, instance=0):
        out = np.zeros((self.N + 1, self.nv))
        if name.startswith("new_"):
            capi.check(self.lib.idocp_unparnmpc_get_new_solution(self.h, name[4:].encode(), instance, P(out)), "get_new_solution")
        elif name[0] == "d":
            capi.check(self.lib.idocp_unocp_get_direction(self.h, name.encode(), instance, P(out)), "get_direction")
        else:
            capi.check(self.lib.idocp_unocp_get_solution(self.h, name.encode(), instance, P(out)), "get_solution")
        return out[:self.N]

def anymal_model():
    return capi.model_from_urdf(ANYMAL_URDF, ANYMAL_CONTACT_FRAMES)

ANYMAL_Q_STANDING = np.array([0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0])

def anymal_problem(model, trotting_ref=True, cone="linearized"):
    """Cost / constraints of examples/anymal/anymal_trotting.cpp:33-107:
    TrottingConfigurationSpaceCost + ContactForceCost, 6 joint limits + LinearizedFrictionCone(mu=0.7).
    cone="nonlinear": FrictionCone / ImpulseFrictionCone instead (examples/anymal/ocp_benchmark.cpp:76)."""
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", ANYMAL_Q_STANDING)
    cost.set("q_weight", np.full(nv, 10.0)).set("qf_weight", np.full(nv, 10.0))
    vw = np.concatenate([np.ones(6), np.full(12, 0.1)])
    aw = np.concatenate([np.full(6, 0.1), np.full(12, 0.01)])
    cost.set("v_weight", vw).set("vf_weight", vw).set("a_weight", aw)
    w = -model.total_mass * model.gravity[2]
    for c in range(4):
        for k in range(3):
            cost.f_weight[c][k] = 0.001
            cost.f_ref[c][k] = 0.0
        cost.f_ref[c][2] = 0.25 * w                      # ContactForceCost::set_f_ref(robot)
    if trotting_ref:
        cost.use_trotting_ref = 1
        cost.t_start, cost.t_period, cost.step_length = 0.5, 0.5, 0.15
        cost.front_swing_knee, cost.hip_swing_knee = 1.7, 1.7
        cost.front_stance_knee, cost.hip_stance_knee = 0.0, 0.0
    # impulse-stage weights of the same example (:74-82, :93): qi = q, vi = v, dvi = a, fi = f weights
    cost.set("qi_weight", np.full(nv, 10.0)).set("vi_weight", vw).set("dvi_weight", aw)
    for c in range(4):
        for k in range(3):
            cost.fi_weight[c][k] = 0.001
            cost.fi_ref[c][k] = 0.0
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    if cone == "nonlinear":
        cons.friction_cone = 1
        cons.impulse_friction_cone = 1
    else:
        cons.linearized_friction_cone = 1
        cons.linearized_impulse_friction_cone = 1
    cons.mu = 0.7
    return cost, cons

def task_space_problem(model, dim=6, frame_id=22, weight=1000.0, time_varying=False):
    """Cost / constraints of examples/iiwa14/task_space_ocp.cpp:54-76 as data: ConfigurationSpaceCost with zero position weights and
    0.01 on v, a (u weights zero), a TaskSpace6DCost (dim=3: TaskSpace3DCost) of weight 1000 on the end-effector frame (pinocchio
    frame 22 = iiwa_link_ee_kuka) with the reference pose of the example's circle at t = 0; six joint limits.
    Returns (cost, cons); time_varying=True leaves the per-stage references to `task_circle_refs`."""
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", np.zeros(nv))
    cost.set("q_weight", np.zeros(nv)).set("qf_weight", np.zeros(nv))
    cost.set("v_weight", np.full(nv, 0.01)).set("vf_weight", np.full(nv, 0.01)).set("a_weight", np.full(nv, 0.01))
    cost.set("u_weight", np.zeros(nv))
    joint = C.c_int()
    R, p = (C.c_double * 9)(), (C.c_double * 3)()
    capi.check(capi.lib().idocp_model_frame_placement(IIWA_URDF.encode(), frame_id, C.byref(joint), R, p), "frame_placement")
    cost.task_dim = dim
    cost.task_joint = joint.value
    for k in range(9):
        cost.task_frame_R[k] = R[k]
    for k in range(3):
        cost.task_frame_p[k] = p[k]
    for k in range(6):
        cost.task_weight[k] = weight
        cost.task_weightf[k] = weight
    ref = task_circle_refs(0.0, 0.0, 0)[0]
    for k in range(12):
        cost.task_ref[k] = ref[k]
    cost.task_time_varying = 1 if time_varying else 0
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    return cost, cons


def task_circle_refs(t, dt, N):
    """TimeVaryingTaskSpace6DRef of examples/iiwa14/task_space_ocp.cpp:21-46 at the stage times t + i dt, i = 0 .. N:
    rotation [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], position (0.546, 0.1 sin(pi t), 0.76 + 0.1 cos(pi t)).  [N + 1][12]"""
    refs = np.zeros((N + 1, 12))
    for i in range(N + 1):
        ti = t + i * dt
        refs[i, :9] = [0, 0, 1, 0, 1, 0, -1, 0, 0]
        refs[i, 9:] = [0.546, 0.1 * np.sin(np.pi * ti), 0.76 + 0.1 * np.cos(np.pi * ti)]
    return refs


def trotting_sequence(solver, model, num_impulse_phases, t_start=0.5, t_period=0.5, step_length=0.15):
    """Contact sequence of examples/anymal/anymal_trotting.cpp:141-177, transcribed as data: all feet ->
    {LH, RF} at t_start -> {LF, RH} at t_start + t_period -> ... (feet advance by step_length)."""
    pts = anymal_contact_points(model).copy()
    solver.set_contact_status([1, 1, 1, 1], pts)
    solver.push_back_contact_status([0, 1, 1, 0], pts, t_start)
    pts[0, 0] += 0.5 * step_length
    pts[3, 0] += 0.5 * step_length
    solver.push_back_contact_status([1, 0, 0, 1], pts, t_start + t_period)
    for i in range(2, num_impulse_phases + 1):
        if i % 2 == 0:
            pts[1, 0] += step_length
            pts[2, 0] += step_length
            solver.push_back_contact_status([0, 1, 1, 0], pts, t_start + i * t_period)
        else:
            pts[0, 0] += step_length
            pts[3, 0] += step_length
            solver.push_back_contact_status([1, 0, 0, 1], pts, t_start + i * t_period)

ANYMAL_Q_RUNNING_START = ANYMAL_Q_STANDING.copy()
ANYMAL_Q_RUNNING_START[0] = -3.0

def running_problem(model, steps=10):
    """Cost / constraints of examples/anymal/anymal_running.cpp:34-128 (BASELINE.json configs[4]):
    TimeVaryingConfigurationSpaceCost (reference moving with stride / t_period inside the running window) +
    ContactForceCost(f_ref = (0, 0, 70)), six joint limits, linearized (impulse) friction cones with mu = 0.8."""
    nv = model.nv
    stride, t_start = 0.4, 1.0
    t_period = 0.135 + 0.05 + 0.165
    cost = capi.Cost()
    cost.set("q_ref", ANYMAL_Q_RUNNING_START)
    qw = np.concatenate([np.ones(3), np.full(15, 10.0)])
    vw = np.concatenate([np.full(3, 0.01), np.full(15, 0.1)])
    aw = np.full(nv, 0.01)
    cost.set("q_weight", qw).set("qf_weight", qw).set("qi_weight", qw)
    cost.set("v_weight", vw).set("vf_weight", vw).set("vi_weight", vw)
    cost.set("a_weight", aw).set("dvi_weight", aw)
    v_ref = np.zeros(nv)
    v_ref[0] = stride / t_period
    cost.set("v_ref", v_ref)
    cost.use_time_varying_ref = 1
    cost.tv_t_begin, cost.tv_t_end = t_start, t_start + (0.5 + steps) * t_period
    for c in range(4):
        for k, wk in enumerate((1e-1, 1e-1, 1e-7)):
            cost.f_weight[c][k] = wk
            cost.fi_weight[c][k] = wk
            cost.f_ref[c][k] = 0.0
            cost.fi_ref[c][k] = 0.0
        cost.f_ref[c][2] = 70.0
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    cons.linearized_friction_cone = 1
    cons.linearized_impulse_friction_cone = 1
    cons.mu = 0.8
    return cost, cons

def jumping_problem(model):
    """Cost / constraints of examples/anymal/anymal_jumping.cpp:39-111: ConfigurationSpaceCost around the standing posture +
    ContactForceCost(f_ref = (0, 0, 70), weights (1, 1, 0.1)), six joint limits, linearized (impulse) friction cones, mu = 0.7."""
    nv = model.nv
    cost = capi.Cost()
    cost.set("q_ref", ANYMAL_Q_STANDING)
    qw = np.concatenate([np.ones(3), np.full(15, 10.0)])
    vw = np.concatenate([np.full(3, 0.01), np.full(15, 0.1)])
    aw = np.full(nv, 0.01)
    cost.set("q_weight", qw).set("qf_weight", qw).set("qi_weight", qw)
    cost.set("v_weight", vw).set("vf_weight", vw).set("vi_weight", vw)
    cost.set("a_weight", aw).set("dvi_weight", aw)
    for c in range(4):
        for k, wk in enumerate((1.0, 1.0, 0.1)):
            cost.f_weight[c][k] = wk
            cost.fi_weight[c][k] = wk
            cost.f_ref[c][k] = 0.0
            cost.fi_ref[c][k] = 0.0
        cost.f_ref[c][2] = 70.0
    cons = capi.Constraints()
    capi.lib().idocp_constraints_init(C.byref(cons))
    cons.linearized_friction_cone = 1
    cons.linearized_impulse_friction_cone = 1
    cons.mu = 0.7
    return cost, cons


def jumping_sequence(solver, model, jumps=3, jump_length=0.25, t_start=1.0, t_jumping=0.15, t_ground=1.0):
    """Contact sequence of examples/anymal/anymal_jumping.cpp:120-142: all feet on the ground, then per jump a flight phase (no
    contacts; the contact points stay the ones just left) and a landing `jump_length` further on all four feet."""
    pts = anymal_contact_points(model).copy()
    solver.set_contact_status([1, 1, 1, 1], pts)
    events = 0
    for k in range(jumps):
        t_off = t_start + k * (t_jumping + t_ground)
        solver.push_back_contact_status([0, 0, 0, 0], pts, t_off)
        pts = pts.copy()
        pts[:, 0] += jump_length
        solver.push_back_contact_status([1, 1, 1, 1], pts, t_off + t_jumping)
        events += 2
    return events


def running_sequence(solver, model, steps=10):
    """Contact sequence of examples/anymal/anymal_running.cpp:137-215, transcribed as data: all feet -> hind feet {LH, RH}
    -> flight -> front feet {LF, RF} -> hind feet -> ... -> all feet; 3 steps + 2 * (steps + 2) + ... = 6 + 3 steps + 4
    discrete events.  Returns the number of events pushed."""
    stride, hip, t_start = 0.4, 0.2, 1.0
    t_fs, t_fhs, t_hs = 0.135, 0.05, 0.165
    t_period = t_fs + t_fhs + t_hs
    pts = anymal_contact_points(model, ANYMAL_Q_RUNNING_START).copy()
    ALL, HIND, FRONT, NONE = [1, 1, 1, 1], [0, 1, 0, 1], [1, 0, 1, 0], [0, 0, 0, 0]
    n = [0]

    def push(status, t):
        solver.push_back_contact_status(status, pts, t)
        n[0] += 1
    solver.set_contact_status(ALL, pts)
    i_fs, i_fhs, i_hs = 0.125, 0.05, 0.125
    t_initial = i_fs + i_fhs + i_hs
    i_fs2, i_fhs2, i_hs2 = 0.135, 0.055, 0.15
    t_initial2 = i_fs2 + i_fhs2 + i_hs2
    push(HIND, t_start)
    push(NONE, t_start + i_fs)
    pts[[0, 2], 0] += 0.25 * stride
    pts[[1, 3], 0] += 0.25 * stride + 0.5 * hip
    push(FRONT, t_start + i_fs + i_fhs)
    push(HIND, t_start + t_initial)
    push(NONE, t_start + t_initial + i_fs2)
    pts[[0, 2], 0] += 0.5 * stride
    pts[[1, 3], 0] += 0.5 * stride + 0.5 * hip
    push(FRONT, t_start + t_initial + i_fs2 + i_fhs2)
    t_end_init = t_start + t_initial + t_initial2
    for i in range(steps):
        push(HIND, t_end_init + i * t_period)
        push(NONE, t_end_init + i * t_period + t_fs)
        pts[:, 0] += stride
        push(FRONT, t_end_init + i * t_period + t_fs + t_fhs)
    push(HIND, t_end_init + steps * t_period)
    e_fs, e_fhs, e_hs = 0.15, 0.05, 0.15
    push(NONE, t_end_init + steps * t_period + e_fs)
    pts[[0, 2], 0] += 0.5 * stride
    pts[[1, 3], 0] += 0.5 * stride - hip
    push(FRONT, t_end_init + steps * t_period + e_fs + e_fhs)
    push(ALL, t_end_init + steps * t_period + e_fs + e_fhs + e_hs)
    return n[0]

OCP_SOL_FIELDS = {"q": 19, "v": 18, "a": 18, "u": 12, "f": 12, "lmd": 18, "gmm": 18, "beta": 18, "mu": 12, "nu_passive": 6}

OCP_DIR_FIELDS = {"dq": 18, "dv": 18, "da": 18, "du": 12, "df": 12, "dlmd": 18, "dgmm": 18, "dbeta": 18, "dmu": 12,
                  "dnu_passive": 6}

OCP_STAGE_ONLY = ("a", "u", "f", "beta", "mu", "nu_passive", "da", "du", "df", "dbeta", "dmu", "dnu_passive")

OCP_CHAIN_EXTRA = {"xi": 12, "dxi": 12}

NODE_KINDS = ("stage", "impulse", "aux", "lift", "terminal")

def anymal_contact_points(model, q_at=None):
    """World positions of the four feet at q_standing (robot.getContactPoints after
    updateFrameKinematics(q_standing), examples/anymal/anymal_trotting.cpp:141-143): problem set-up, done by the PRODUCT's host
    kinematics (idocp_model_contact_positions) so that neither bench.py nor the GPU tests route their inputs through the
    oracle; tests/test_capi_symbols.py checks that entry against the oracle's frame kinematics."""
    q = arr(ANYMAL_Q_STANDING if q_at is None else q_at)
    pts = np.zeros((model.ncontacts, 3))
    capi.check(capi.lib().idocp_model_contact_positions(C.byref(model), P(q), P(pts)), "idocp_model_contact_positions")
    return pts

class HipOCP:
    """Contact path through the C ABI (idocp_ocp_*)."""
    _post_create = []      # callables applied to every new handle (a test seam: no environment variable changes what production runs)

    def __init__(self, model, cost, cons, T, N, batch=1, device=0, max_num_impulse=0):
        self.lib = capi.lib()
        self.N, self.nv, self.nu, self.nq, self.batch = N, model.nv, model.nu, model.nq, batch
        self.max_events = max_num_impulse
        h = C.c_void_p()
        if max_num_impulse > 0:
            capi.check(self.lib.idocp_ocp_create_hybrid(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse, batch,
                                                        device, C.byref(h)), "idocp_ocp_create_hybrid")
        else:
            capi.check(self.lib.idocp_ocp_create(C.byref(model), C.byref(cost), C.byref(cons), T, N, batch, device, C.byref(h)),
                       "idocp_ocp_create")
        self.h = h
        for hook in type(self)._post_create:      # (tests/helpers.force_forms: kernel forms forced per test; empty in production)
            hook(self)

    # ---- contact sequences with discrete events
    def push_back_contact_status(self, active, points, switching_time):
        a = (C.c_int * 4)(*[int(x) for x in active])
        capi.check(self.lib.idocp_ocp_push_back_contact_status(self.h, a, P(arr(points)), switching_time), "push_back_contact_status")

    def set_contact_points(self, phase, points):
        capi.check(self.lib.idocp_ocp_set_contact_points(self.h, phase, P(arr(points))), "set_contact_points")

    def pop_back_contact_status(self):
        capi.check(self.lib.idocp_ocp_pop_back_contact_status(self.h), "pop_back_contact_status")

    def pop_front_contact_status(self):
        capi.check(self.lib.idocp_ocp_pop_front_contact_status(self.h), "pop_front_contact_status")

    def chain(self, t):
        cap = self.N + 1 + 3 * max(self.max_events, 1)
        IA = lambda: (C.c_int * cap)()
        kind, index, slot, dimf, sw = IA(), IA(), IA(), IA(), IA()
        dt = np.zeros(cap)
        M = self.lib.idocp_ocp_get_chain(self.h, t, cap, kind, index, slot, P(dt), dimf, sw)
        assert M > 0, capi.lib().idocp_last_error()
        return [dict(kind=NODE_KINDS[kind[p]], index=index[p], slot=slot[p], dt=dt[p], dimf=dimf[p], sw_dimi=sw[p]) for p in range(M)]

    def chain_times(self, t):
        """Times of the stages of the chain discretised at t (idocp_ocp_get_chain_times): where a TimeVarying task-space reference is evaluated."""
        cap = self.N + 1 + 3 * max(self.max_events, 1)
        tt = np.zeros(cap)
        M = self.lib.idocp_ocp_get_chain_times(self.h, t, cap, P(tt))
        assert M > 0, capi.lib().idocp_last_error()
        return tt[:M].copy()

    def set_task_refs(self, t, refs):
        """refs[M][12] (rotation row-major, position) for the chain discretised at t (idocp_ocp_set_task_refs)."""
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        capi.check(self.lib.idocp_ocp_set_task_refs(self.h, t, refs.shape[0], P(refs)), "set_task_refs")

    def get_chain(self, name, M, instance=0):
        dim = OCP_SOL_FIELDS.get(name) or OCP_DIR_FIELDS.get(name) or OCP_CHAIN_EXTRA[name]
        out = np.zeros((M, dim))
        fn = self.lib.idocp_ocp_get_solution_chain if (name in OCP_SOL_FIELDS or name == "xi") else self.lib.idocp_ocp_get_direction_chain
        capi.check(fn(self.h, name.encode(), instance, P(out)), "get_chain")
        return out

    def riccati_chain(self, M, instance=0):
        nv, nu = self.nv, self.nu
        Pm, s = np.zeros((M, 2 * nv, 2 * nv)), np.zeros((M, 2 * nv))
        K, k = np.zeros((M - 1, 2 * nv, nu)), np.zeros((M - 1, nu))
        capi.check(self.lib.idocp_ocp_get_riccati_chain(self.h, instance, P(Pm), P(s), P(K), P(k)), "get_riccati_chain")
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.idocp_ocp_destroy(self.h)
            self.h = None

    def set_contact_status(self, active, points):
        a = (C.c_int * 4)(*[int(x) for x in active])
        capi.check(self.lib.idocp_ocp_set_contact_status_uniformly(self.h, a, P(arr(points))), "set_contact_status")

    def set_solution(self, name, value):
        capi.check(self.lib.idocp_ocp_set_solution(self.h, name.encode(), P(arr(value))), "set_solution")

    def set_solution_batch(self, name, values):
        capi.check(self.lib.idocp_ocp_set_solution_batch(self.h, name.encode(), P(arr(values))), "set_solution_batch")

    def init_constraints(self, t=0.0):
        capi.check(self.lib.idocp_ocp_init_constraints(self.h, t), "init_constraints")

    def _bc(self, x, dim):
        x = arr(x)
        return arr(np.broadcast_to(x, (self.batch, dim))) if x.ndim == 1 else x

    def update(self, t, q, v):
        return self.lib.idocp_ocp_update_solution(self.h, t, P(self._bc(q, self.nq)), P(self._bc(v, self.nv)), 0)

    def infeasible_stage(self):
        ok, where = np.zeros(self.batch, dtype=np.int32), np.zeros(self.batch, dtype=np.int32)
        capi.check(self.lib.idocp_ocp_is_current_solution_feasible(self.h, ok.ctypes.data_as(capi.c_int_p),
                                                                   where.ctypes.data_as(capi.c_int_p)), "is_feasible")
        assert np.all((where < 0) == (ok == 1))
        return where

    def kkt_error(self, t, q, v):
        capi.check(self.lib.idocp_ocp_compute_kkt_residual(self.h, t, P(self._bc(q, self.nq)), P(self._bc(v, self.nv))),
                   "compute_kkt_residual")
        out = np.zeros(self.batch)
        capi.check(self.lib.idocp_ocp_kkt_error(self.h, P(out)), "kkt_error")
        return out

    def get(self, name, instance=0):
        if name in OCP_SOL_FIELDS:
            dim, fn = OCP_SOL_FIELDS[name], self.lib.idocp_ocp_get_solution
        else:
            dim, fn = OCP_DIR_FIELDS[name], self.lib.idocp_ocp_get_direction
        out = np.zeros((self.N + 1, dim))
        capi.check(fn(self.h, name.encode(), instance, P(out)), "get " + name)
        return out[:self.N] if name in OCP_STAGE_ONLY else out

    def step_sizes(self):
        a, b = np.zeros(self.batch), np.zeros(self.batch)
        capi.check(self.lib.idocp_ocp_get_step_sizes(self.h, P(a), P(b)), "get_step_sizes")
        return a, b

    def riccati(self, instance=0):
        nv, nu, N = self.nv, self.nu, self.N
        Pm, s = np.zeros((N + 1, 2 * nv, 2 * nv)), np.zeros((N + 1, 2 * nv))
        K, k = np.zeros((N, 2 * nv, nu)), np.zeros((N, nu))
        capi.check(self.lib.idocp_ocp_get_riccati(self.h, instance, P(Pm), P(s), P(K), P(k)), "get_riccati")
        return Pm.transpose(0, 2, 1), s, K.transpose(0, 2, 1), k

    def constraint_data(self, instance=0):
        dimc = self.lib.idocp_ocp_dimc(self.h)
        sl, du = np.zeros((self.N, dimc)), np.zeros((self.N, dimc))
        capi.check(self.lib.idocp_ocp_get_constraint_data(self.h, instance, P(sl), P(du)), "get_constraint_data")
        return sl, du

    def lqr_stage(self, i, instance=0):
        nv, nu = self.nv, self.nu
        nx = 2 * nv
        Qxx, Qxu, Quu, A, B = np.zeros((nx, nx)), np.zeros((nu, nx)), np.zeros((nu, nu)), np.zeros((nx, nx)), np.zeros((nu, nx))
        lx, lu, Fx = np.zeros(nx), np.zeros(nu), np.zeros(nx)
        capi.check(self.lib.idocp_ocp_get_lqr_stage(self.h, instance, i, P(Qxx), P(Qxu), P(Quu), P(A), P(B), P(lx), P(lu), P(Fx)),
                   "get_lqr_stage")
        return Qxx.T, Qxu.T, Quu.T, A.T, B.T, lx, lu, Fx

class HipParNMPC(HipOCP):
    """ParNMPCSolver through the C ABI (idocp_parnmpc_* + the shared idocp_ocp_* entry points)."""

    def set_stage_values(self, name, values):
        """warm start: one field of stages 0 .. len(values) - 1, all instances"""
        values = arr(values)
        capi.check(self.lib.idocp_ocp_set_solution_stages(self.h, name.encode(), values.shape[0], P(values)), "set_solution_stages " + name)

    def set_aux_mats(self, mats):
        """warm start: aux_mat [stages, nx, nx] (row, col)"""
        cm = arr(np.asarray(mats).transpose(0, 2, 1))      # column-major per stage
        capi.check(self.lib.idocp_parnmpc_set_aux_mat(self.h, cm.shape[0], P(cm)), "set_aux_mat")

    def set_chain_values(self, name, values):
        """warm start along the CHAIN of the current discretisation (event stages included): values[M, dim]"""
        values = arr(values)
        capi.check(self.lib.idocp_ocp_set_solution_chain(self.h, name.encode(), values.shape[0], P(values)), "set_solution_chain " + name)

    def set_chain_aux_mats(self, mats):
        cm = arr(np.asarray(mats).transpose(0, 2, 1))
        capi.check(self.lib.idocp_parnmpc_set_aux_mat_chain(self.h, cm.shape[0], P(cm)), "set_aux_mat_chain")

    def __init__(self, model, cost, cons, T, N, batch=1, device=0, max_num_impulse=0):
        self.lib = capi.lib()
        self.N, self.nv, self.nu, self.nq, self.batch = N, model.nv, model.nu, model.nq, batch
        self.max_events = max_num_impulse
        h = C.c_void_p()
        if max_num_impulse > 0:
            capi.check(self.lib.idocp_parnmpc_create_hybrid(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse, batch, device,
                                                            C.byref(h)), "idocp_parnmpc_create_hybrid")
        else:
            capi.check(self.lib.idocp_parnmpc_create(C.byref(model), C.byref(cost), C.byref(cons), T, N, batch, device, C.byref(h)),
                       "idocp_parnmpc_create")
        self.h = h

    def init(self, t=0.0):
        capi.check(self.lib.idocp_parnmpc_init_backward_correction(self.h, t), "init_backward_correction")
        self.init_constraints(t)

    def update(self, t, q, v):
        return self.lib.idocp_parnmpc_update_solution(self.h, t, P(self._bc(q, self.nq)), P(self._bc(v, self.nv)), 0)

    def kkt_error(self, t, q, v):
        capi.check(self.lib.idocp_parnmpc_compute_kkt_residual(self.h, t, P(self._bc(q, self.nq)), P(self._bc(v, self.nv))),
                   "compute_kkt_residual")
        out = np.zeros(self.batch)
        capi.check(self.lib.idocp_ocp_kkt_error(self.h, P(out)), "kkt_error")
        return out

    def get(self, name, instance=0):
        if name in OCP_SOL_FIELDS:
            dim, fn = OCP_SOL_FIELDS[name], self.lib.idocp_ocp_get_solution
        else:
            dim, fn = OCP_DIR_FIELDS[name], self.lib.idocp_ocp_get_direction
        out = np.zeros((self.N, dim))
        capi.check(fn(self.h, name.encode(), instance, P(out)), "get " + name)
        return out

def warm_start_parnmpc(ocp, targets, N):
    """The well-posed N = 256 workload of BASELINE configs[3]: ParNMPC started, like in an MPC loop, from the converged Riccati
    solution of the same problem -- stage i of ParNMPC lives at the time of grid stage i + 1 of the OCP: (q, v, lmd, gmm) and
    aux_mat = P of stage i + 1, (a, u, f, beta, mu) of stage min(i + 1, N - 1).  From the reference's cold start (aux_mat = terminal
    Hessian everywhere) the forward correction sweep amplifies by 1.15 per stage and the first direction reaches 3e12 at N = 256;
    from here it is O(1) and the iteration converges.  `ocp`: a converged OracleOCP / HipOCP; `targets`: ParNMPC solvers."""
    sol = {f: np.asarray(ocp.get(f)) for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu")}
    Pm = ocp.riccati()[0]
    vals = {f: sol[f][1:N + 1] for f in ("q", "v", "lmd", "gmm")}
    for f in ("a", "u", "f", "beta", "mu"):
        vals[f] = np.stack([sol[f][min(i + 1, N - 1)] for i in range(N)])
    aux = Pm[1:N + 1]
    for t in targets:
        for f, v in vals.items():
            t.set_stage_values(f, v)
        t.set_aux_mats(aux)


def map_ocp_onto_parnmpc_chain(co, cp, sol, Pm):
    """Warm start of ParNMPC on a horizon WITH discrete events from the converged OCPSolver solution of the same problem (the state an
    MPC loop that switches solvers is in).  co / cp: the chains of the OCP (forward Euler: stage, [impulse, aux | lift], stage, ...,
    terminal) and of ParNMPC (backward Euler: [aux, impulse | lift] IN FRONT of the grid stage behind the event; without its
    placeholder); sol: the OCP's solution fields along its chain; Pm: its Riccati matrices P along the chain.  Node by node in time:
    ParNMPC's aux stage (state just before the impulse) <- the OCP's impulse stage, its impulse stage (state just after) <- the OCP's
    aux stage with dv, f, beta, mu of the OCP's impulse stage, a lift stage <- the lift stage, a grid stage <- the OCP node at (or
    right behind) the same time; (a, u, f, beta, mu) from an OCP node with the same number of active feet; aux_mat = P of the state's
    node.  Returns (values: dict field -> [Mp, dim], aux: [Mp, nx, nx])."""
    def times(ch, backward):
        t, out = 0.0, []
        for c in ch:
            d = 0.0 if c["kind"] == "impulse" else c["dt"]
            if backward:
                t += d; out.append(t)
            else:
                out.append(t); t += d
        return np.array(out)
    Mo = len(co)
    to, tp = times(co, False), times(cp, True)
    idx = {(c["kind"], c["index"]): p for p, c in enumerate(co)}
    state_src, dyn_src = [], []
    for p, c in enumerate(cp):
        k = c["kind"]
        if k == "aux":
            s_ = idx[("impulse", c["index"])]; d_ = max(s_ - 1, 0)
        elif k == "impulse":
            s_ = idx[("aux", c["index"])]; d_ = idx[("impulse", c["index"])]
        elif k == "lift":
            s_ = idx[("lift", c["index"])]; d_ = s_
        else:
            cand = [pp for pp, cc in enumerate(co) if cc["kind"] in ("stage", "aux", "lift", "terminal") and to[pp] >= tp[p] - 1e-9]
            s_ = cand[0] if cand else Mo - 1
            d_ = min(s_, Mo - 2)
            while co[d_]["dimf"] != c["dimf"] or co[d_]["kind"] == "impulse":
                d_ -= 1
        state_src.append(s_); dyn_src.append(d_)
    vals = {f: np.stack([sol[f][s_] for s_ in state_src]) for f in ("q", "v", "lmd", "gmm")}
    for f in ("a", "u", "f", "beta", "mu"):
        vals[f] = np.stack([sol[f][d_] for d_ in dyn_src])
    aux = np.stack([Pm[s_] for s_ in state_src])
    return vals, aux


class ParNMPCShardHandle:
    """One shard of a ParNMPC horizon on this process's GPU through the C ABI: the grid stages [rank N / world, (rank + 1) N / world)
    (idocp_parnmpc_create_shard; with max_num_impulse > 0 idocp_parnmpc_create_hybrid_shard: every rank holds the whole contact
    sequence and keeps its slice of the chain).  Driven by the library's C++ driver (idocp_parnmpc_dist_*)."""

    def __init__(self, model, cost, cons, T, N, rank, world, batch, device, max_num_impulse=0):
        assert N % world == 0, "the horizon must divide evenly among the ranks"
        self.lib = capi.lib()
        self.Nl, self.batch, self.rank, self.world, self.N = N // world, batch, rank, world, N
        self.nq, self.nv = model.nq, model.nv
        self.max_events = max_num_impulse
        h = C.c_void_p()
        if max_num_impulse > 0:
            capi.check(self.lib.idocp_parnmpc_create_hybrid_shard(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse,
                                                                  rank * self.Nl, (rank + 1) * self.Nl, batch, device, C.byref(h)),
                       "idocp_parnmpc_create_hybrid_shard")
        else:
            capi.check(self.lib.idocp_parnmpc_create_shard(C.byref(model), C.byref(cost), C.byref(cons), T / world, self.Nl, rank * self.Nl,
                                                           1 if rank == world - 1 else 0, 1 if rank > 0 else 0, batch, device, C.byref(h)),
                       "idocp_parnmpc_create_shard")
        self.h = h
        dq, dv = C.c_void_p(), C.c_void_p()
        capi.check(self.lib.idocp_parnmpc_prev_state(self.h, C.byref(dq), C.byref(dv)))
        self.d_q, self.d_v = dq, dv

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.idocp_ocp_destroy(self.h)
            self.h = None

    # the contact-sequence interface of the solvers (trotting_sequence)
    def set_contact_status(self, active, points):
        capi.check(self.lib.idocp_ocp_set_contact_status_uniformly(self.h, (C.c_int * 4)(*[int(x) for x in active]), P(arr(points))))

    def push_back_contact_status(self, active, points, t_ev):
        capi.check(self.lib.idocp_ocp_push_back_contact_status(self.h, (C.c_int * 4)(*[int(x) for x in active]), P(arr(points)), t_ev))

    def set_solution(self, name, value):
        capi.check(self.lib.idocp_ocp_set_solution(self.h, name.encode(), P(arr(value))))

    def chain(self, t=0.0):
        cap = self.N + 1 + 3 * max(self.max_events, 1)
        IA = lambda: (C.c_int * cap)()
        kind, index, slot, dimf, sw = IA(), IA(), IA(), IA(), IA()
        dt = np.zeros(cap)
        M = self.lib.idocp_ocp_get_chain(self.h, t, cap, kind, index, slot, P(dt), dimf, sw)
        assert M > 0, capi.lib().idocp_last_error()
        return [dict(kind=NODE_KINDS[kind[p]], index=index[p], slot=slot[p], dt=dt[p], dimf=dimf[p]) for p in range(M)]


def trotting_parnmpc_by_continuation(model, T, N, device=0, K=30, log=None):
    """A CONVERGED, MOVING trotting solution of ParNMPC at BASELINE configs[3]'s size, by continuation (round 4).  ParNMPC has no
    globalisation and its undamped iteration does not contract from a standing start on the trotting problem; it does from the converged
    OCPSolver solution of the trot IN PLACE (step length 0, swing-knee reference 0: forward and backward Euler nearly coincide on a
    solution that hardly moves), and from there the step length and the knee reference are raised to the example's 0.15 m / 1.7 rad in
    small steps, re-converging from the previous solution each time (a failed step is retried at half the increment from a clone of the
    last converged solver).  Returns (cost of the full problem, solver, chain length without the placeholder): a HipParNMPC of batch 1
    holding the converged solution of the full problem."""
    n_events = int((T - 0.5125) / 0.5) + 1
    lib = capi.lib()
    lib.idocp_ocp_set_cost.argtypes = [C.c_void_p, C.POINTER(capi.Cost)]
    lib.idocp_ocp_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]

    def problem(lam):
        cost, cons = anymal_problem(model, trotting_ref=True)
        cost.step_length = 0.15 * lam
        cost.front_swing_knee = cost.hip_swing_knee = 1.7 * lam
        return cost, cons

    def points(lam):
        class Rec:
            def __init__(self):
                self.pts = []

            def set_contact_status(self, a, p):
                self.pts.append(np.array(p).copy())

            def push_back_contact_status(self, a, p, t):
                self.pts.append(np.array(p).copy())
        r = Rec()
        trotting_sequence(r, model, n_events - 1, t_start=0.5125, step_length=0.15 * lam)
        return r.pts

    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(model.nv)
    cost0, cons = problem(0.0)
    ocp = HipOCP(model, cost0, cons, T, N, batch=1, device=device, max_num_impulse=n_events)
    pn = HipParNMPC(model, cost0, cons, T, N, batch=1, device=device, max_num_impulse=n_events)
    for s in (ocp, pn):
        trotting_sequence(s, model, n_events - 1, t_start=0.5125, step_length=0.0)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
    ocp.init_constraints(0.0)
    pn.init(0.0)
    for it in range(80):
        assert ocp.update(0.0, q, v) == 0
        if it > 10 and ocp.kkt_error(0.0, q, v)[0] < 1e-9:
            break
    co, cp = ocp.chain(0.0), pn.chain(0.0)[:-1]
    Mo, Mp = len(co), len(cp)
    sol = {f: ocp.get_chain(f, Mo) for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu")}
    vals, aux = map_ocp_onto_parnmpc_chain(co, cp, sol, ocp.riccati_chain(Mo)[0])
    for f, x in vals.items():
        pn.set_chain_values(f, x)
    pn.set_chain_aux_mats(aux)
    pn.init_constraints(0.0)
    del ocp

    def converge(maxit=40, tol=1e-6):
        e = float("inf")
        for it in range(maxit):
            r = lib.idocp_parnmpc_update_solution(pn.h, 0.0, P(pn._bc(q, pn.nq)), P(pn._bc(v, pn.nv)), 0)
            e = float(pn.kkt_error(0.0, q, v)[0])
            if r or not np.isfinite(e) or e > 1e6:
                return False, it + 1, e
            if e < tol:
                return True, it + 1, e
        return False, maxit, e

    ok, its, e = converge()
    if log is not None:
        log.append(dict(lam=0.0, ok=ok, iterations=its, kkt=e))
    assert ok, "ParNMPC does not converge on the trot in place (KKT %.3e)" % e
    lam, step, tries = 0.0, 1.0 / K, 0
    while lam < 1.0 - 1e-12:
        tries += 1
        assert tries < 600 and step > 1e-5, "the continuation stalls at lam = %.4f" % lam
        lam_try = min(1.0, lam + step)
        ck = C.c_void_p()
        capi.check(lib.idocp_ocp_clone(pn.h, C.byref(ck)), "clone")
        cost, _ = problem(lam_try)
        capi.check(lib.idocp_ocp_set_cost(pn.h, C.byref(cost)), "set_cost")
        for ph, pts in enumerate(points(lam_try)):
            capi.check(lib.idocp_ocp_set_contact_points(pn.h, ph, P(arr(pts))), "set_contact_points")
        ok, its, e = converge()
        if log is not None:
            log.append(dict(lam=lam_try, step=step, ok=ok, iterations=its, kkt=e))
        if ok:
            lib.idocp_ocp_destroy(ck)
            lam, step = lam_try, min(1.0 / K, step * 1.5)
        else:
            lib.idocp_ocp_destroy(pn.h)
            pn.h = ck
            step *= 0.5
    return problem(1.0)[0], pn, Mp
