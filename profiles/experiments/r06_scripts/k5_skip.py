import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idocp_amd import capi
from idocp_amd.workloads import ANYMAL_Q_STANDING, HipOCP, anymal_model, anymal_problem, trotting_sequence
m = anymal_model(); cost, cons = anymal_problem(m, trotting_ref=True)
nimp = 9; N = 100; T = 0.5 + nimp * 0.5 + 0.05; B = 1024
g = HipOCP(m, cost, cons, T, N, batch=B, max_num_impulse=nimp + 1)
trotting_sequence(g, m, nimp)
q0 = np.tile(ANYMAL_Q_STANDING, (B, 1)); v0 = np.zeros((B, m.nv))
g.set_solution("q", ANYMAL_Q_STANDING); g.set_solution("v", np.zeros(m.nv)); g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
g.init_constraints(0.0)
g.update(0.0, q0, v0)
lib = capi.lib()
rt = C.CDLL("libamdhip64.so")
d_q, d_v = C.c_void_p(), C.c_void_p()
capi.check(lib.idocp_device_alloc(C.byref(d_q), q0.nbytes)); capi.check(lib.idocp_device_alloc(C.byref(d_v), v0.nbytes))
capi.check(lib.idocp_device_upload(d_q, q0.ctypes.data, q0.nbytes)); capi.check(lib.idocp_device_upload(d_v, v0.ctypes.data, v0.nbytes))
for _ in range(3): capi.check(lib.idocp_ocp_launch_kernel(g.h, 8, d_q, d_v))
capi.check(lib.idocp_ocp_synchronize(g.h))
t0 = time.perf_counter()
n = 20
for _ in range(n): capi.check(lib.idocp_ocp_launch_kernel(g.h, 8, d_q, d_v))
capi.check(lib.idocp_ocp_synchronize(g.h))
print(os.environ.get("IDOCP_HIP_LIB"), "K5 ms %.3f" % (1e3 * (time.perf_counter() - t0) / n))
