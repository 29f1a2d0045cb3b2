"""ParNMPC N = 256: one handle of 256 instances against two handles of 128 on streams of their own, stepped alternately (their serial sweeps can
run beside the other handle's stage-parallel kernels)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idocp_amd import capi
from idocp_amd.workloads import ANYMAL_Q_STANDING, HipParNMPC, anymal_contact_points, anymal_model, anymal_problem
m = anymal_model(); cost, cons = anymal_problem(m, trotting_ref=True); pts = anymal_contact_points(m)
N, T = 256, 12.8
lib = capi.lib()
def make(B):
    g = HipParNMPC(m, cost, cons, T, N, batch=B)
    g.set_contact_status([1, 1, 1, 1], pts)
    g.set_solution("q", ANYMAL_Q_STANDING); g.set_solution("v", np.zeros(m.nv)); g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    g.init(0.0)
    q = np.tile(ANYMAL_Q_STANDING, (B, 1)); v = np.zeros((B, m.nv))
    dq, dv = C.c_void_p(), C.c_void_p()
    capi.check(lib.idocp_device_alloc(C.byref(dq), q.nbytes)); capi.check(lib.idocp_device_alloc(C.byref(dv), v.nbytes))
    capi.check(lib.idocp_device_upload(dq, q.ctypes.data, q.nbytes)); capi.check(lib.idocp_device_upload(dv, v.ctypes.data, v.nbytes))
    return g, dq, dv
def run(handles, steps=10):
    fn = lib.idocp_parnmpc_update_solution_device
    for _ in range(2):
        for g, dq, dv in handles: capi.check(fn(g.h, 0.0, dq, dv), "update")
    for g, _, _ in handles: capi.check(lib.idocp_ocp_synchronize(g.h))
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, dq, dv in handles: capi.check(fn(g.h, 0.0, dq, dv), "update")
    for g, _, _ in handles: capi.check(lib.idocp_ocp_synchronize(g.h))
    return 1e3 * (time.perf_counter() - t0) / steps
lib.idocp_parnmpc_update_solution_device.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
one = [make(256)]
print("one handle of 256: %.3f ms per iteration of 256 instances" % run(one))
del one
two = [make(128), make(128)]
print("two handles of 128: %.3f ms per iteration of 256 instances" % run(two))
del two
four = [make(64) for _ in range(4)]
print("four handles of 64: %.3f ms per iteration of 256 instances" % run(four))
