"""The RCCL transport of the sharded ParNMPC driver (idocp_amd/csrc/parnmpc_dist.hip) on the ONE GPU a test box has.

north_star: "shards horizon stages across the 8 GPUs ... RCCL halo exchange over xGMI"; counterpart of
/root/reference/src/ocp/backward_correction_solver.cpp:255-366.  More than one rank cannot run here, so these tests run everything a
rank does against RCCL itself with world = 1: dlopen + symbol table, ncclGetUniqueId (128 bytes), ncclCommInitRank, grouped
ncclSend / ncclRecv of every halo kind (to this very rank), ncclAllReduce (sum, min) and ncclBroadcast on the shard's own stream
between its kernels -- the world-1 early-outs of the driver are switched off (idocp_comm_set_force_collectives)."""
import ctypes as C

import numpy as np
import pytest

from helpers import ANYMAL_Q_STANDING, P, anymal_contact_points, anymal_problem, arr, rel_err
from test_parnmpc_gpu import make_pair

pytestmark = pytest.mark.gpu


def _shard_and_comm(m, cost, cons, T, N, batch=1):
    from idocp_amd import capi
    from parnmpc_dist import HipParNMPCShard
    lib = capi.lib()
    sh = HipParNMPCShard(m, cost, cons, T, N, 0, 1, batch, 0)
    raw = (C.c_char * 128)()
    capi.check(lib.idocp_comm_get_unique_id(raw), "comm_get_unique_id")
    assert any(raw.raw), "ncclGetUniqueId left the id empty"
    comm = C.c_void_p()
    capi.check(lib.idocp_comm_init_rank(raw, 0, 1, 0, C.byref(comm)), "comm_init_rank")
    assert lib.idocp_comm_rank(comm) == 0 and lib.idocp_comm_world(comm) == 1
    # what RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion): the figures bench.py prints as config.rccl
    lib.idocp_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    nr, ur, ver, tr = C.c_int(-1), C.c_int(-1), C.c_int(0), C.c_int(-1)
    capi.check(lib.idocp_comm_info(comm, C.byref(nr), C.byref(ur), C.byref(ver), C.byref(tr)), "comm_info")
    assert (nr.value, ur.value, tr.value) == (1, 0, 1) and ver.value > 20000, (nr.value, ur.value, ver.value, tr.value)
    return lib, sh, comm


def test_rccl_send_recv_allreduce_broadcast_to_self():
    from idocp_amd import capi
    m, o, g, q, v = make_pair(8, 0.4)
    cost, cons = anymal_problem(m, trotting_ref=False)
    lib, sh, comm = _shard_and_comm(m, cost, cons, 0.4, 8, batch=3)
    capi.check(lib.idocp_parnmpc_dist_attach(sh.h, comm), "attach")
    # a second attach must be refused (it would leak the first set of halo buffers)
    assert lib.idocp_parnmpc_dist_attach(sh.h, comm) != 0
    worst = C.c_double(-1.0)
    capi.check(lib.idocp_parnmpc_dist_transport_selftest(sh.h, C.byref(worst)), "transport_selftest")
    assert worst.value == 0.0, worst.value
    # wrong dimensions are refused before anything is copied
    assert lib.idocp_parnmpc_dist_set_initial_state(sh.h, P(arr(np.zeros((3, 19)))), P(arr(np.zeros((3, 18)))), 18, 18) != 0
    # destroying an attached handle detaches it (no stale entry for a later handle at the same address)
    lib.idocp_ocp_destroy(sh.h)
    sh.h = None
    assert lib.idocp_parnmpc_dist_detach(None) != 0
    lib.idocp_comm_destroy(comm)


def test_driver_through_rccl_equals_the_plain_solver():
    """idocp_parnmpc_dist_update_solution / _kkt_error on a world-1 RCCL communicator with the collectives FORCED through RCCL
    (all-reduce of the step sizes and of the squared KKT error, broadcast of the terminal aux_mat) against the plain single-handle
    solver over four iterations: the transport must not change a bit of the iterate."""
    from idocp_amd import capi
    m, o, g, q, v = make_pair(12, 0.6)
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    lib, sh, comm = _shard_and_comm(m, cost, cons, 0.6, 12)
    capi.check(lib.idocp_comm_set_force_collectives(comm, 1))
    capi.check(lib.idocp_ocp_set_contact_status_uniformly(sh.h, (C.c_int * 4)(1, 1, 1, 1), P(arr(pts))))
    capi.check(lib.idocp_ocp_set_solution(sh.h, b"q", P(arr(ANYMAL_Q_STANDING))))
    capi.check(lib.idocp_ocp_set_solution(sh.h, b"v", P(np.zeros(m.nv))))
    capi.check(lib.idocp_ocp_set_solution(sh.h, b"f", P(arr([0, 0, 0.25 * (-m.total_mass * m.gravity[2])]))))
    capi.check(lib.idocp_parnmpc_dist_attach(sh.h, comm), "attach")
    capi.check(lib.idocp_parnmpc_dist_set_initial_state(sh.h, P(arr(q[None, :])), P(arr(v[None, :])), m.nq, m.nv))
    capi.check(lib.idocp_parnmpc_dist_init_backward_correction(sh.h, 0.0), "init")
    capi.check(lib.idocp_ocp_init_constraints(sh.h, 0.0))

    def get(name, dim):
        out = np.zeros((12, dim))
        capi.check(lib.idocp_ocp_get_solution(sh.h, name.encode(), 0, P(out)))
        return out

    for it in range(4):
        assert g.update(0.0, q, v) == 0
        capi.check(lib.idocp_parnmpc_dist_update_solution(sh.h, 0.0), "update")
        capi.check(lib.idocp_ocp_synchronize(sh.h))
        for name, dim in (("q", 19), ("v", 18), ("u", 12), ("lmd", 18), ("a", 18), ("f", 12)):
            assert rel_err(get(name, dim), g.get(name)) < 1e-12, (it, name)
    kkt = np.zeros(1)
    capi.check(lib.idocp_parnmpc_dist_kkt_error(sh.h, 0.0, P(kkt)), "kkt")
    e_g = g.kkt_error(0.0, q, v)[0]
    assert abs(kkt[0] - e_g) < 1e-10 * max(1.0, e_g)
    capi.check(lib.idocp_parnmpc_dist_detach(sh.h))
    lib.idocp_comm_destroy(comm)
