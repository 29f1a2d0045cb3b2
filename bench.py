#!/usr/bin/env python3
"""bench.py -- SQP iterations/s of the idocp hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line from rank 0.  A "step" is one updateSolution() of every OCP
instance of the batch, i.e. one pass of the hot path (K1 linearize -> S1/S2
Riccati -> K2 expand -> K3 integrate) over the synthetic batch; the reference's
protocol is ocpbenchmarker::CPUTime (include/idocp/utils/ocp_benchmarker.hxx:
13-34): repeated updateSolution(t, q, v) at fixed (t, q, v).

Multi-GPU: the Riccati path does not shard along the horizon, so N
ranks run N independent replicas of the batch ("replicas only", weak scaling);
the only collective is the barrier / max-reduction of the timing.

Inputs (q, v, the solver state) are resident in HBM when the timed region
starts.  Per-kernel durations are measured with HIP events recorded on the
handle's stream between the kernel launches of every timed step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
A_STAGE_IIWA = 5544            # algorithmic bytes per stage, SURVEY.md 8(d) / DESIGN.md 5
KERNELS = ["un_linearize", "un_riccati_backward", "un_riccati_forward", "un_expand", "un_reduce_steps", "un_integrate"]


class Hip:
    """The handful of HIP runtime calls bench.py needs (events on our own stream)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [C.c_void_p]
        self.rt.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.rt.hipDeviceSynchronize.argtypes = []
        self.rt.hipSetDevice.argtypes = [C.c_int]

    def event(self):
        e = C.c_void_p()
        assert self.rt.hipEventCreate(C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.rt.hipEventRecord(e, stream) == 0

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        assert self.rt.hipEventSynchronize(b) == 0
        assert self.rt.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value


def cpu_baseline(model, cost, cons, T, N, q, v, target_seconds=12.0):
    """CPU restatement (oracle, kind "port") timed on this host with the
    reference's CPUTime protocol, single thread, on a bounded sample."""
    from helpers import OracleUnOCP, P, arr, oracle
    o = OracleUnOCP(model, cost, cons, T, N)
    o.set_solution("q", q)
    o.set_solution("v", v)
    lib = oracle()
    ric = C.c_double()
    for _ in range(50):                       # converge first (examples/iiwa14/unocp_benchmark.cpp:50-52)
        o.update(0.0, q, v)
    t_probe = lib.oracle_unocp_bench(o.h, 0.0, P(arr(q)), P(arr(v)), 20, C.byref(ric))
    iters = int(max(20, min(20000, target_seconds / max(t_probe / 20, 1e-9))))
    el = lib.oracle_unocp_bench(o.h, 0.0, P(arr(q)), P(arr(v)), iters, C.byref(ric))
    return {"value": iters / el, "unit": "SQP iterations/s", "cores": 1, "kind": "port",
            "sample": "%d updateSolution calls of one iiwa14 N=%d instance, single thread (oracle/, -O3)" % (iters, N),
            "ms_per_update": 1e3 * el / iters, "ms_per_riccati_sweep": 1e3 * ric.value / iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16384, help="independent OCP instances per GPU")
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from idocp_amd import capi
    from helpers import HipUnOCP, iiwa14_model, unocp_problem
    lib = capi.lib()                       # fails loudly if the HIP extension is missing
    hip = Hip()
    hip.rt.hipSetDevice(local_rank)

    # workload: BASELINE.json configs[1] -- iiwa14 UnOCPSolver, N=100, T=5 (dt=0.05), FP64,
    # batch of instances with perturbed initial states
    N, T, B = args.horizon, 0.05 * args.horizon, args.batch
    model = iiwa14_model()
    cost, cons = unocp_problem(model)
    nv = model.nv
    rng = np.random.default_rng(20240 + rank)
    q0 = np.ascontiguousarray(2.0 + 0.1 * rng.uniform(-1, 1, (B, nv)))
    v0 = np.zeros((B, nv))
    solver = HipUnOCP(model, cost, cons, T, N, batch=B, device=local_rank)
    solver.set_solution_batch("q", q0)
    solver.set_solution("v", v0[0])
    d_q, d_v = C.c_void_p(), C.c_void_p()
    capi.check(lib.idocp_device_alloc(C.byref(d_q), q0.nbytes))
    capi.check(lib.idocp_device_alloc(C.byref(d_v), v0.nbytes))
    capi.check(lib.idocp_device_upload(d_q, q0.ctypes.data, q0.nbytes))
    capi.check(lib.idocp_device_upload(d_v, v0.ctypes.data, v0.nbytes))
    stream = lib.idocp_unocp_stream(solver.h)

    def step(events=None):
        for kid in range(len(KERNELS)):
            if events is not None:
                hip.record(events[kid], stream)
            capi.check(lib.idocp_unocp_launch_kernel(solver.h, kid, d_q, d_v), KERNELS[kid])
        if events is not None:
            hip.record(events[len(KERNELS)], stream)

    def sync():
        capi.check(lib.idocp_unocp_synchronize(solver.h))
        hip.rt.hipDeviceSynchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ev = [[hip.event() for _ in range(len(KERNELS) + 1)] for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    sync()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())

    # per-kernel average durations from the events of the timed region
    kms = np.zeros(len(KERNELS))
    for k in range(args.steps):
        for kid in range(len(KERNELS)):
            kms[kid] += hip.elapsed_ms(ev[k][kid], ev[k][kid + 1])
    kms /= args.steps
    dom = int(np.argmax(kms))
    units = {0: B * N, 1: B * N, 2: B * N, 3: B * (N + 1), 4: B * N, 5: B * (N + 1)}
    alg_bytes = A_STAGE_IIWA * units[dom]
    achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("batch") == B and rec.get("horizon") == N:
                traffic = rec.get("hbm_bytes_per_launch", {}).get(KERNELS[dom])
        except Exception:
            traffic = None

    # parity guard inside the bench: the timed state must still be a valid solver state
    kkt = solver.kkt_error(0.0, q0, v0)
    assert np.isfinite(kkt).all(), "non-finite KKT error after the timed region"

    if rank == 0:
        total_iters = world * B * args.steps
        ms_step = 1e3 * el / args.steps
        out = {
            "metric": "SQP iterations/sec (whole node)", "value": total_iters / el, "unit": "SQP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "iiwa14 UnOCPSolver N=%d T=%.2f FP64 (BASELINE.json configs[1]); "
                                   "batch=%d independent OCP instances per GPU, replicas across GPUs" % (N, T, B),
                       "horizon": N, "batch_per_gpu": B, "parallelism": "replicas x%d" % world,
                       "ms_per_riccati_sweep": float(kms[1] + kms[2]),
                       "kernel_ms": {KERNELS[i]: float(kms[i]) for i in range(len(KERNELS))},
                       "max_kkt_error_after": float(np.max(kkt))},
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": float(kms[dom]),
                         "whole_step_frac": A_STAGE_IIWA * B * (N + 1) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, cost, cons, T, N, q0[0], v0[0])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
