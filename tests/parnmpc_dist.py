"""TEST SCAFFOLDING (round 4: moved out of the product package).  The multi-GPU driver of the product is C++
(idocp_amd/csrc/parnmpc_dist.hip, idocp_parnmpc_dist_*); this Python restatement of the same halo protocol exists so that the
protocol can be exercised without GPUs -- on gloo, with the CPU oracle as the shard backend (tests/test_parnmpc_dist.py) -- and so
that single phases of the HIP shards can be driven by hand in a test (tests/test_parnmpc_gpu.py, test_parnmpc_hybrid_gpu.py).

Horizon-sharded ParNMPC: one process per shard of the horizon, halo exchange between neighbours.

ParNMPC (src/ocp/parnmpc_solver.cpp:73-103) is stage-parallel except for two thin serial sweeps (the backward correction
of (lmd, gmm) and the forward correction of (q, v)), so the stages [r N/G, (r+1) N/G) of one horizon can live on rank r
(BASELINE.json configs[3]: N = 256 over 2 / 4 / 8 GPUs).  Per iteration a rank exchanges with its neighbours only

  state_last   (q, v)            of its last stage   -> right   (the "previous state" of the neighbour's first stage)
  costate_first(lmd, gmm, q)     of its first stage  -> left    (coupling terms of the neighbour's last stage)
  aux_first    aux_mat           of its first stage  -> left    (added to the neighbour's last Qxx)
  bwd_first    corrected lmd,gmm of its first stage  -> left    (pipeline of the backward serial sweep)
  fwd_last     corrected q, v    of its last stage   -> right   (pipeline of the forward serial sweep)

plus an all-reduce(min) of the two step sizes and an all-reduce(sum) of the squared KKT error.  The collectives are
torch.distributed point-to-point sends over RCCL (xGMI) on the GPUs; the same driver runs on gloo with the CPU oracle
as the shard backend (tests/test_parnmpc_dist.py), which is how the protocol is tested without GPUs.

A shard backend provides:
  batch, halo_size(kind), export(kind) -> tensor[batch, size], import_(kind, tensor), phase(name, t),
  local_steps() -> tensor[batch, 2], set_steps(tensor), err2(t) -> tensor[batch]
"""
STATE_LAST, COSTATE_FIRST, AUX_FIRST, BWD_FIRST, FWD_LAST, AUX_ALL = 0, 1, 2, 3, 4, 5


class ShardedParNMPC:
    def __init__(self, shard, dist, rank, world, max_step=None):
        self.s, self.dist, self.rank, self.world, self.max_step = shard, dist, rank, world, max_step
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None

    # ---- point-to-point helpers (a send and a receive with different peers may be in flight together)
    def _sendrecv(self, send_kind, send_to, recv_kind, recv_from):
        import torch
        reqs = []
        rbuf = None
        if send_to is not None:
            reqs.append(self.dist.isend(self.s.export(send_kind).contiguous(), dst=send_to))
        if recv_from is not None:
            like = self.s.export(recv_kind)
            rbuf = torch.empty_like(like)
            reqs.append(self.dist.irecv(rbuf, src=recv_from))
        for r in reqs:
            r.wait()
        return rbuf

    def init_backward_correction(self, t):
        """ParNMPCSolver::initBackwardCorrection: aux_mat = terminal cost Hessian at the LAST stage of the horizon, so the
        last rank computes it and everybody takes its value."""
        import torch
        self.s.phase("init_aux", t)
        buf = self.s.export(AUX_ALL).contiguous()
        self.dist.broadcast(buf, src=self.world - 1)
        self.s.import_(AUX_ALL, buf)

    def exchange_boundary(self):
        got = self._sendrecv(STATE_LAST, self.right, STATE_LAST, self.left)
        if got is not None:
            self.s.import_(STATE_LAST, got)
        got = self._sendrecv(COSTATE_FIRST, self.left, COSTATE_FIRST, self.right)
        if got is not None:
            self.s.import_(COSTATE_FIRST, got)
        got = self._sendrecv(AUX_FIRST, self.left, AUX_FIRST, self.right)
        if got is not None:
            self.s.import_(AUX_FIRST, got)

    def update(self, t):
        import torch
        self.exchange_boundary()
        self.s.phase("linearize", t)
        # backward serial sweep: right to left
        if self.right is not None:
            buf = torch.empty_like(self.s.export(BWD_FIRST))
            self.dist.recv(buf, src=self.right)
            self.s.import_(BWD_FIRST, buf)
        self.s.phase("bwd_serial", t)
        if self.left is not None:
            self.dist.send(self.s.export(BWD_FIRST).contiguous(), dst=self.left)
        self.s.phase("bwd_parallel", t)
        # forward serial sweep: left to right
        if self.left is not None:
            buf = torch.empty_like(self.s.export(FWD_LAST))
            self.dist.recv(buf, src=self.left)
            self.s.import_(FWD_LAST, buf)
        self.s.phase("fwd_serial", t)
        if self.right is not None:
            self.dist.send(self.s.export(FWD_LAST).contiguous(), dst=self.right)
        self.s.phase("fwd_parallel", t)
        steps = self.s.local_steps().contiguous()
        if self.world > 1:
            self.dist.all_reduce(steps, op=self.dist.ReduceOp.MIN)
        if self.max_step is not None:               # damped iteration (ParNMPC itself has no globalisation)
            steps = steps.clamp(max=self.max_step)
        self.s.set_steps(steps)
        self.s.phase("integrate", t)

    def kkt_error(self, t):
        self.exchange_boundary()
        e2 = self.s.err2(t).contiguous()
        if self.world > 1:
            self.dist.all_reduce(e2, op=self.dist.ReduceOp.SUM)
        return e2.sqrt()


class HipParNMPCShard:
    """Shard backend on the HIP path: the stages [rank N/world, (rank+1) N/world) of the horizon on this process's GPU.
    Halos travel as torch CUDA tensors whose device pointers are handed to the C ABI (include/idocp_hip.h)."""
    PHASES = {"linearize": (0, 1, 2), "bwd_serial": (3,), "bwd_parallel": (4,), "fwd_serial": (5,), "fwd_parallel": (6, 7, 8),
              "integrate": (9,)}

    def __init__(self, model, cost, cons, T, N, rank, world, batch, device, max_num_impulse=0):
        import ctypes as C
        import torch
        from idocp_amd import capi
        assert N % world == 0, "the horizon must divide evenly among the ranks"
        self.C, self.torch, self.capi = C, torch, capi
        self.lib = capi.lib()
        self.Nl, self.batch, self.rank, self.world = N // world, batch, rank, world
        self.dev = torch.device("cuda", device)
        h = C.c_void_p()
        if max_num_impulse > 0:
            # a horizon with discrete events: every rank holds the whole contact sequence and keeps its slice of the chain
            capi.check(self.lib.idocp_parnmpc_create_hybrid_shard(C.byref(model), C.byref(cost), C.byref(cons), T, N, max_num_impulse,
                                                                  rank * self.Nl, (rank + 1) * self.Nl, batch, device, C.byref(h)),
                       "idocp_parnmpc_create_hybrid_shard")
        else:
            capi.check(self.lib.idocp_parnmpc_create_shard(C.byref(model), C.byref(cost), C.byref(cons), T / world, self.Nl, rank * self.Nl,
                                                           1 if rank == world - 1 else 0, 1 if rank > 0 else 0, batch, device, C.byref(h)),
                       "idocp_parnmpc_create_shard")
        self.h = h
        dq, dv, ds = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(self.lib.idocp_parnmpc_prev_state(self.h, C.byref(dq), C.byref(dv)))
        capi.check(self.lib.idocp_parnmpc_step_sizes_device(self.h, C.byref(ds)))
        self.d_q, self.d_v, self.d_steps = dq, dv, ds

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.idocp_ocp_destroy(self.h)
            self.h = None

    def set_initial_state(self, q, v):
        """rank 0: the measured state q[batch][nq], v[batch][nv] (host arrays)"""
        import numpy as np
        q, v = np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        self.capi.check(self.lib.idocp_device_upload(self.d_q, q.ctypes.data, q.nbytes))
        self.capi.check(self.lib.idocp_device_upload(self.d_v, v.ctypes.data, v.nbytes))

    def halo_size(self, kind):
        return self.lib.idocp_parnmpc_halo_size(kind)

    def export(self, kind):
        buf = self.torch.empty((self.batch, self.halo_size(kind)), dtype=self.torch.float64, device=self.dev)
        self.capi.check(self.lib.idocp_parnmpc_export_halo(self.h, kind, buf.data_ptr()), "export_halo")
        return buf

    def import_(self, kind, tensor):
        tensor = tensor.contiguous()
        self.torch.cuda.synchronize(self.dev)
        self.capi.check(self.lib.idocp_parnmpc_import_halo(self.h, kind, tensor.data_ptr()), "import_halo")

    def phase(self, name, t):
        if name == "init_aux":
            self.capi.check(self.lib.idocp_parnmpc_init_backward_correction(self.h, t), "init_backward_correction")
            return
        if name == "linearize":
            self.capi.check(self.lib.idocp_parnmpc_discretize(self.h, t), "discretize")
        for ph in self.PHASES[name]:
            self.capi.check(self.lib.idocp_parnmpc_launch_phase(self.h, ph, self.d_q, self.d_v), "phase %d" % ph)
        self.capi.check(self.lib.idocp_ocp_synchronize(self.h))

    def local_steps(self):
        buf = self.torch.empty((self.batch, 2), dtype=self.torch.float64, device=self.dev)
        self.capi.check(self.lib.idocp_device_copy(buf.data_ptr(), self.d_steps, buf.numel() * 8))
        return buf

    def set_steps(self, tensor):
        tensor = tensor.contiguous()
        self.torch.cuda.synchronize(self.dev)
        self.capi.check(self.lib.idocp_device_copy(self.d_steps, tensor.data_ptr(), tensor.numel() * 8))

    def err2(self, t):
        buf = self.torch.empty((self.batch,), dtype=self.torch.float64, device=self.dev)
        self.capi.check(self.lib.idocp_parnmpc_kkt_error_squared_device(self.h, t, buf.data_ptr()), "kkt_error_squared")
        return buf


class HipUnParNMPCShard:
    """Shard backend of the fixed-base UnParNMPCSolver on the HIP path (idocp_unparnmpc_create_shard): the same protocol
    through ShardedParNMPC.  aux_mat is initialised locally on every rank (the terminal cost Hessian of the fixed-base cost is
    constant), so the AUX_ALL broadcast carries nothing."""
    PHASES = {"linearize": (0, 1), "bwd_serial": (2,), "bwd_parallel": (3,), "fwd_serial": (4,), "fwd_parallel": (5,), "integrate": (6,)}

    def __init__(self, model, cost, cons, T, N, rank, world, batch, device):
        import ctypes as C
        import torch
        from idocp_amd import capi
        assert N % world == 0, "the horizon must divide evenly among the ranks"
        self.C, self.torch, self.capi = C, torch, capi
        self.lib = capi.lib()
        self.Nl, self.batch, self.rank, self.world = N // world, batch, rank, world
        self.dev = torch.device("cuda", device)
        h = C.c_void_p()
        capi.check(self.lib.idocp_unparnmpc_create_shard(C.byref(model), C.byref(cost), C.byref(cons), T, N, rank * self.Nl,
                                                         (rank + 1) * self.Nl, batch, device, C.byref(h)), "idocp_unparnmpc_create_shard")
        self.h = h
        dq, dv, ds = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(self.lib.idocp_unparnmpc_prev_state(self.h, C.byref(dq), C.byref(dv)))
        capi.check(self.lib.idocp_unparnmpc_step_sizes_device(self.h, C.byref(ds)))
        self.d_q, self.d_v, self.d_steps = dq, dv, ds

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.idocp_unocp_destroy(self.h)
            self.h = None

    def set_initial_state(self, q, v):
        """rank 0: the measured state q[batch][nv], v[batch][nv] (host arrays)"""
        import numpy as np
        q, v = np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        self.capi.check(self.lib.idocp_device_upload(self.d_q, q.ctypes.data, q.nbytes))
        self.capi.check(self.lib.idocp_device_upload(self.d_v, v.ctypes.data, v.nbytes))

    def halo_size(self, kind):
        return 1 if kind == AUX_ALL else self.lib.idocp_unparnmpc_halo_size(kind)

    def export(self, kind):
        buf = self.torch.zeros((self.batch, self.halo_size(kind)), dtype=self.torch.float64, device=self.dev)
        if kind != AUX_ALL:
            self.capi.check(self.lib.idocp_unparnmpc_export_halo(self.h, kind, buf.data_ptr()), "export_halo")
        return buf

    def import_(self, kind, tensor):
        if kind == AUX_ALL:
            return
        tensor = tensor.contiguous()
        self.torch.cuda.synchronize(self.dev)
        self.capi.check(self.lib.idocp_unparnmpc_import_halo(self.h, kind, tensor.data_ptr()), "import_halo")

    def phase(self, name, t):
        if name == "init_aux":
            self.capi.check(self.lib.idocp_unocp_init_constraints(self.h), "init_constraints")
            self.capi.check(self.lib.idocp_unparnmpc_init_backward_correction(self.h, t), "init_backward_correction")
            return
        for ph in self.PHASES[name]:
            self.capi.check(self.lib.idocp_unparnmpc_launch_phase(self.h, ph, self.d_q, self.d_v), "phase %d" % ph)
        self.capi.check(self.lib.idocp_unocp_synchronize(self.h))

    def local_steps(self):
        buf = self.torch.empty((self.batch, 2), dtype=self.torch.float64, device=self.dev)
        self.capi.check(self.lib.idocp_device_copy(buf.data_ptr(), self.d_steps, buf.numel() * 8))
        return buf

    def set_steps(self, tensor):
        tensor = tensor.contiguous()
        self.torch.cuda.synchronize(self.dev)
        self.capi.check(self.lib.idocp_device_copy(self.d_steps, tensor.data_ptr(), tensor.numel() * 8))

    def err2(self, t):
        buf = self.torch.empty((self.batch,), dtype=self.torch.float64, device=self.dev)
        self.capi.check(self.lib.idocp_unparnmpc_kkt_error_squared_device(self.h, t, buf.data_ptr()), "kkt_error_squared")
        return buf
