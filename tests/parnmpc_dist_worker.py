#!/usr/bin/env python3
"""TEST SCAFFOLDING: one RANK of the product's C++ sharded ParNMPC driver (idocp_amd/csrc/parnmpc_dist.hip, idocp_parnmpc_dist_*) as a process
of its own -- started by tests/test_parnmpc_gpu.py::test_cxx_driver_two_processes... BEFORE anything in this process touches the GPU.

    parnmpc_dist_worker.py <rank> <world> <port> <N> <T> <iters> <outdir>

The ranks share the one GPU of the box, where RCCL cannot connect them (it refuses two ranks on one device), so the driver runs on its third
transport: idocp_comm_init_callbacks, host-staged, backed here by torch.distributed / gloo.  The DRIVER is the real one: its grouping
(ncclGroupStart / End -> group_start / group_end), the order of its sends and receives, the pipelines of the two serial sweeps across ranks, the
all-reduces, and the lifetime of communicator and shard (attach, detach, idocp_comm_destroy, idocp_ocp_destroy in a process that exits cleanly).
Every transport call is logged; the rank writes its stages' solution after every iteration and the log to <outdir>/rank<r>.npz."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    T, iters, outdir = float(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        torch.cuda.init()      # torch's HIP runtime first, the library binds to the same one (tests/conftest.py)

    from helpers import ANYMAL_Q_STANDING, P, anymal_contact_points, anymal_model, anymal_problem, arr
    from idocp_amd import capi
    from parnmpc_dist import HipParNMPCShard
    lib = capi.lib()

    log, reqs, keep = [], [], []

    def view(buf, n):
        return torch.from_numpy(np.ctypeslib.as_array(buf, shape=(n,)))

    SENDF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_ulong, C.c_int, C.c_int)
    GRPF = C.CFUNCTYPE(C.c_int, C.c_void_p)
    REDF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_ulong, C.c_int)
    DESF = C.CFUNCTYPE(None, C.c_void_p)

    def guarded(fn):
        def wrapped(*a):
            try:
                return fn(*a)
            except Exception as e:      # noqa: BLE001  (an exception must not unwind through the C caller)
                sys.stderr.write("rank %d: transport callback failed: %r\n" % (rank, e))
                return 1
        return wrapped

    @guarded
    def send(_ctx, buf, n, peer, in_group):
        log.append(("send", peer, int(n), int(in_group)))
        t = view(buf, n)
        if in_group:
            keep.append(t)
            reqs.append(dist.isend(t, dst=peer))
        else:
            dist.send(t, dst=peer)
        return 0

    @guarded
    def recv(_ctx, buf, n, peer, in_group):
        log.append(("recv", peer, int(n), int(in_group)))
        t = view(buf, n)
        if in_group:
            keep.append(t)
            reqs.append(dist.irecv(t, src=peer))
        else:
            dist.recv(t, src=peer)
        return 0

    @guarded
    def group_start(_ctx):
        log.append(("group_start", -1, 0, 0))
        return 0

    @guarded
    def group_end(_ctx):
        log.append(("group_end", -1, len(reqs), 0))
        for r in reqs:
            r.wait()
        del reqs[:]
        del keep[:]
        return 0

    @guarded
    def allreduce(_ctx, buf, n, op):
        log.append(("allreduce", -1, int(n), int(op)))
        dist.all_reduce(view(buf, n), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)
        return 0

    @guarded
    def broadcast(_ctx, buf, n, root):
        log.append(("broadcast", root, int(n), 0))
        dist.broadcast(view(buf, n), src=root)
        return 0

    destroyed = []

    class Callbacks(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("send", SENDF), ("recv", SENDF), ("group_start", GRPF), ("group_end", GRPF), ("allreduce", REDF),
                    ("broadcast", REDF), ("destroy", DESF)]
    cb = Callbacks(None, SENDF(send), SENDF(recv), GRPF(group_start), GRPF(group_end), REDF(allreduce), REDF(broadcast), DESF(lambda _c: destroyed.append(1)))

    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    pts = anymal_contact_points(m)
    Nl = N // world
    shard = HipParNMPCShard(m, cost, cons, T, N, rank, world, 1, 0)
    comm = C.c_void_p()
    lib.idocp_comm_init_callbacks.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Callbacks), C.POINTER(C.c_void_p)]
    capi.check(lib.idocp_comm_init_callbacks(rank, world, 0, C.byref(cb), C.byref(comm)), "comm_init_callbacks")
    a = (C.c_int * 4)(1, 1, 1, 1)
    capi.check(lib.idocp_ocp_set_contact_status_uniformly(shard.h, a, P(arr(pts))))
    capi.check(lib.idocp_ocp_set_solution(shard.h, b"q", P(arr(ANYMAL_Q_STANDING))))
    capi.check(lib.idocp_ocp_set_solution(shard.h, b"v", P(np.zeros(m.nv))))
    capi.check(lib.idocp_ocp_set_solution(shard.h, b"f", P(arr([0, 0, 0.25 * (-m.total_mass * m.gravity[2])]))))
    capi.check(lib.idocp_parnmpc_dist_attach(shard.h, comm), "attach")
    q = ANYMAL_Q_STANDING.copy()
    q[7:] += 0.05
    v = np.zeros(m.nv)
    if rank == 0:
        capi.check(lib.idocp_parnmpc_dist_set_initial_state(shard.h, P(arr(q[None, :])), P(arr(v[None, :])), m.nq, m.nv))
    lib.idocp_parnmpc_dist_transport_selftest.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    dev = C.c_double(-1.0)
    capi.check(lib.idocp_parnmpc_dist_transport_selftest(shard.h, C.byref(dev)), "selftest")
    n_selftest = len(log)
    capi.check(lib.idocp_parnmpc_dist_init_backward_correction(shard.h, 0.0), "init_backward_correction")
    capi.check(lib.idocp_ocp_init_constraints(shard.h, 0.0))
    out = {"selftest": np.array([dev.value])}
    marks = [len(log)]
    for it in range(iters):
        capi.check(lib.idocp_parnmpc_dist_update_solution(shard.h, 0.0), "dist_update_solution")
        capi.check(lib.idocp_ocp_synchronize(shard.h))
        marks.append(len(log))
        for name, dim in (("q", 19), ("v", 18), ("u", 12), ("lmd", 18), ("a", 18), ("f", 12)):
            x = np.zeros((Nl, dim))
            capi.check(lib.idocp_ocp_get_solution(shard.h, name.encode(), 0, P(x)))
            out["it%d_%s" % (it, name)] = x
    kkt = np.zeros(1)
    capi.check(lib.idocp_parnmpc_dist_kkt_error(shard.h, 0.0, P(kkt)), "dist_kkt_error")
    out["kkt"] = kkt
    # lifetime: detach, destroy the communicator (its destroy callback fires), destroy the shard; the process then exits cleanly
    capi.check(lib.idocp_parnmpc_dist_detach(shard.h))
    lib.idocp_comm_destroy(comm)
    assert destroyed == [1]
    lib.idocp_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    out["log_op"] = np.array([e[0] for e in log])
    out["log_args"] = np.array([e[1:] for e in log], dtype=np.int64)
    out["marks"] = np.array([n_selftest] + marks)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    del shard
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
