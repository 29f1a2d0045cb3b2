"""CSTAMP / RSTAMP of S3 (build -DIDOCP_S3_STAMPS): the phases of a stage with a switching constraint and of a regular stage, batch 1024."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from idocp_amd import capi
from idocp_amd.workloads import ANYMAL_Q_STANDING, HipOCP, anymal_model, anymal_problem, trotting_sequence
m = anymal_model(); cost, cons = anymal_problem(m, trotting_ref=True)
nimp = 9; N = 100; T = 0.5 + nimp * 0.5 + 0.05; B = 1024
g = HipOCP(m, cost, cons, T, N, batch=B, max_num_impulse=nimp + 1)
trotting_sequence(g, m, nimp)
rng = np.random.default_rng(1)
q0 = np.tile(ANYMAL_Q_STANDING, (B, 1)); q0[:, 7:] += 0.01 * rng.uniform(-1, 1, (B, 12)); v0 = 0.01 * rng.uniform(-1, 1, (B, m.nv))
g.set_solution_batch("q", q0); g.set_solution("v", np.zeros(m.nv)); g.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
g.init_constraints(0.0)
for _ in range(3): assert g.update(0.0, q0, v0) == 0
lib = capi.lib()
out = (C.c_longlong * 64)()
lib.idocp_ocp_get_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
assert lib.idocp_ocp_get_profile(g.h, out, 64) == 0
s = np.array(out[:], dtype=np.int64)
print("RSTAMP (regular stage, us from stamp 0):", " ".join("%d:%.2f" % (k, (s[16 + k] - s[16]) / 100.0) for k in range(16) if s[16 + k] > 0))
print("CSTAMP (constrained stage):", " ".join("%d:%.2f" % (k, (s[32 + k] - s[32]) / 100.0) for k in range(16) if s[32 + k] > 0))
