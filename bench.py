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
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector / matrix spec peak (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
# algorithmic bytes per stage, SURVEY.md 8(d) / DESIGN.md 4: A_stage = 8 [n_s + 2 n_c + n_d + 2 n_c + 2 (n_P + n_sP) + n_K] at the LARGEST stage
# class of the workload (ANYmal: 12 contact rows); the per-class figures are a_stage_of() below
A_STAGE = {"iiwa14": 5544, "iiwa14_task_space": 5544, "iiwa14_unparnmpc": 5656, "anymal": 25032, "anymal_trotting": 25032, "anymal_running": 25032, "anymal_parnmpc": 25032,
           "anymal_parnmpc_trotting": 25032}


def a_stage_of(kind, dimf, sw_dimi=0, nv=18, nu=12):
    """SURVEY 8(d)'s formula evaluated for ONE stage of an ANYmal chain by its class (the survey quotes the nf = 12 stage, 25 032 B):
    n_s solution doubles, n_c inequality rows, n_d direction doubles, P / s hand-off written and read once, gain written.
      grid / aux / lift stage with nf contact rows: n_s = 127 + 2 nf (+ xi), n_c = 72 + 5 nf / 3, n_d = 126 + 2 nf (+ dxi), n_K = nu (2 nv + 1)
      impulse stage (no u, no nu_passive, impulse friction cone only): n_s = 109 + 2 nf, n_c = 5 nf / 3, n_d = 108 + 2 nf, n_K = 0
      terminal stage: n_s = 4 nv + 1, n_d = 4 nv, nothing else but P, s"""
    n_P, n_sP = 3 * nv * nv, 2 * nv
    if kind == "terminal":
        n_s, n_c, n_d, n_K = 4 * nv + 1, 0, 4 * nv, 0
    elif kind == "impulse":
        n_s, n_c, n_d, n_K = 6 * nv + 1 + 2 * dimf, 5 * dimf // 3, 6 * nv + 2 * dimf, 0
    else:
        n_s, n_c, n_d, n_K = 6 * nv + 1 + nu + 6 + 2 * dimf + sw_dimi, 6 * nu + 5 * dimf // 3, 6 * nv + nu + 6 + 2 * dimf + sw_dimi, nu * (2 * nv + 1)
    return 8 * (n_s + 2 * n_c + n_d + 2 * n_c + 2 * (n_P + n_sP) + n_K)


def oracle_flops_record(workload, chain_stages):
    """FLOPs of one SQP iteration of ONE instance by region, from tests/golden/oracle_flops.json -- DATA: the exact operation counts of the
    CPU restatement taken by its counting build (oracle/flops.hpp, tests/golden/gen_oracle_flops.py; tests/test_oracle_flops.py holds the
    file to a live recount).  None when the file has no entry for this workload at this chain length."""
    try:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_flops.json")))
    except (OSError, ValueError):
        return None
    w = rec.get("workloads", {}).get({"iiwa14_task_space": "iiwa14"}.get(workload, workload))
    if not w or w.get("chain_stages") != chain_stages:
        return None
    return {"regions": {n: sum(r.values()) for n, r in w["regions"].items() if n != "other"}, "kernel_regions": rec["kernel_regions"],
            "flop_per_iteration": w["flop_per_iteration"]}


def kernel_flops(rec, kernel):
    """FLOPs per instance and iteration of the regions a kernel of the product executes (kernel_regions of the record)"""
    alias = {"ocp_nominal": "ocp_nominal+ocp_rnea(switch)", "ocp_rnea": None, "parnmpc_backward_serial": "parnmpc_corrections"}
    key = alias.get(kernel, kernel)
    regs = rec["kernel_regions"].get(key) if key else None
    if not regs:
        return None
    return sum(rec["regions"].get(r, 0) for r in regs)


def quoted_sq(workload, batch, horizon, kernel):
    """Issue counters of `kernel` from the newest committed SQ record of this workload (profiles/rNN_pmc_sq_<workload>.json, written by
    profiles/make_sq_json.py from the rocprofv3 --pmc SQ_* passes of profiles/run_profiles.sh ... sq) -- a QUOTATION under the same rule
    as quoted_traffic: same workload / batch / horizon, the kernel's own name, not older than the newest kernel trace."""
    import glob
    import re
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_%s.json" % workload)), reverse=True)
    if not recs:
        return None, "no committed SQ counter record for this workload"
    rnd = lambda path: int(re.match(r"r(\d+)_", os.path.basename(path)).group(1))
    traces = glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_trace.txt" % workload))
    try:
        rec = json.load(open(recs[0]))
    except Exception as e:
        return None, "unreadable record %s (%s)" % (os.path.basename(recs[0]), e)
    name = os.path.basename(recs[0])
    if rec.get("batch") != batch or rec.get("horizon") != horizon:
        return None, "profiles/%s was taken at batch %s / horizon %s, this run is %d / %d" % (name, rec.get("batch"), rec.get("horizon"), batch, horizon)
    if rnd(recs[0]) < max([rnd(t) for t in traces], default=0):
        return None, "profiles/%s predates the newest kernel trace of this workload" % name
    val = rec.get("kernels", {}).get(kernel)
    if val is None:
        return None, "profiles/%s has no kernel named %s" % (name, kernel)
    return val, "quoted from profiles/%s (round %s; rocprofv3 --pmc SQ_* passes of this command; not measured in this run)" % (name, rec.get("round"))


def quoted_traffic(workload, batch, horizon, kernel):
    """HBM bytes per launch of `kernel` from the NEWEST committed counter record of this workload (profiles/rNN_pmc_traffic_<workload>.json:
    rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in passes of their own, gfx950 correction applied, profiles/run_profiles.sh) -- a QUOTATION,
    labelled as one, and only when it describes what this run measures: same workload / batch / horizon, the record lists that very kernel
    (keys are the kernels' own names: no fallback to a similarly named one), and no newer kernel-trace profile of the workload exists
    whose round the record predates (a record of an earlier round describes earlier kernels).  Otherwise (None, reason)."""
    import glob
    import re
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % workload)), reverse=True)
    if not recs:
        return None, "no committed counter record for this workload"
    rnd = lambda path: int(re.match(r"r(\d+)_", os.path.basename(path)).group(1))
    traces = glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_trace.txt" % workload))
    newest_trace = max([rnd(t) for t in traces], default=0)
    try:
        rec = json.load(open(recs[0]))
    except Exception as e:
        return None, "unreadable record %s (%s)" % (os.path.basename(recs[0]), e)
    name = os.path.basename(recs[0])
    if rec.get("workload", workload) != workload or rec.get("batch") != batch or rec.get("horizon") != horizon:
        return None, "profiles/%s was taken at batch %s / horizon %s, this run is %d / %d" % (name, rec.get("batch"), rec.get("horizon"), batch, horizon)
    if rnd(recs[0]) < newest_trace:
        return None, "profiles/%s predates the kernels of the round-%d kernel trace of this workload" % (name, newest_trace)
    val = rec.get("hbm_bytes_per_launch", {}).get(kernel)
    if val is None:
        return None, "profiles/%s has no kernel named %s" % (name, kernel)
    return val, ("quoted from profiles/%s (round %s; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel %s; not measured in this run)"
                 % (name, rec.get("round"), kernel))
KERNELS_UN = ["un_linearize", "un_riccati_backward", "un_riccati_forward", "un_expand", "un_reduce_steps", "un_integrate"]
KERNELS_UNP = ["un_linearize", "unparnmpc_coarse_update", "unparnmpc_backward_serial", "unparnmpc_backward_parallel",
               "unparnmpc_forward_serial", "unparnmpc_expand", "un_integrate"]
KERNELS_OCP = ["ocp_rnea", "ocp_condense", "ocp_riccati_backward", "ocp_riccati_forward", "ocp_expand_primal",
               "ocp_reduce_steps", "ocp_expand_dual_integrate"]
# the same step with the condensation phase bracketed in its two halves (idocp_ocp_launch_kernel ids 7, 8): the nominal rigid-body
# sweeps are a kernel of their own (ocp_nominal_kernel), and the roofline line prices ocp_condense_kernel's launches alone, as the
# rocprofv3 summary under profiles/ lists them
KERNELS_OCP_SPLIT = ["ocp_rnea", "ocp_nominal", "ocp_condense", "ocp_riccati_backward", "ocp_riccati_forward", "ocp_expand_primal",
                     "ocp_reduce_steps", "ocp_expand_dual_integrate"]
KIDS_OCP_SPLIT = [0, 7, 8, 2, 3, 4, 5, 6]
# round 5: the forward sweep expands as it walks (ocp_forward_expand_kernel = S4 + K6 + the step-size reduction behind kernel id 3;
# idocp_ocp_fused_forward(h) == 1: batches of instances).  IDOCP_FUSED_FORWARD=0 brings the three kernels of the lists above back.
KERNELS_OCP_FUSED = ["ocp_rnea", "ocp_nominal", "ocp_condense", "ocp_riccati_backward", "ocp_forward_expand", "ocp_expand_dual_integrate"]
KIDS_OCP_FUSED = [0, 7, 8, 2, 3, 6]


def ocp_kernel_lists(lib, h, B, Mc):
    """(names, kernel ids, units per launch by position in the list, positions of the backward / forward sweep) of one OCPSolver iteration"""
    lib.idocp_ocp_fused_forward.argtypes = [C.c_void_p]
    if lib.idocp_ocp_fused_forward(h):
        return KERNELS_OCP_FUSED, KIDS_OCP_FUSED, {0: B * (Mc - 1), 1: B * Mc, 2: B * Mc, 3: B * (Mc - 1), 4: B * Mc, 5: B * Mc}, (3, 4)
    return (KERNELS_OCP_SPLIT, KIDS_OCP_SPLIT,
            {0: B * (Mc - 1), 1: B * Mc, 2: B * Mc, 3: B * (Mc - 1), 4: B * (Mc - 1), 5: B * Mc, 6: B * (Mc - 1), 7: B * Mc}, (3, 4))


class Hip:
    """The handful of HIP runtime calls bench.py needs (events on our own stream)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [C.c_void_p]
        self.rt.hipEventDestroy.argtypes = [C.c_void_p]
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


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def native_oracle():
    """The CPU baseline library: oracle/ built ON THIS HOST with -O3 -march=native -fopenmp (oracle/Makefile, target
    liboracle_native.so; a few seconds of g++).  Falls back to the portable build (-march=x86-64-v3 -fopenmp) that travelled with
    the tree when there is no compiler.  Returns (path, flags)."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir, "liboracle_native.so"], capture_output=True, text=True)
    path = os.path.join(odir, "liboracle_native.so")
    if r.returncode == 0 and os.path.exists(path):
        return path, "-O3 -march=native -fopenmp"
    return os.path.join(odir, "liboracle.so"), "-O3 -march=x86-64-v3 -fopenmp"


def cpu_baseline(workload, model, cost, cons, T, N, q, v, pts=None, target_seconds=6.0, nimp=0):
    """CPU restatement (oracle, kind "port") timed on this host with the reference's CPUTime protocol
    (ocp_benchmarker.hxx:13-34: repeated updateSolution at fixed (t, q, v) after convergence) on a bounded sample, with the
    reference's OpenMP stage loops on nthreads = 1, 4 and all cores.  `value` is the best row."""
    import helpers
    from helpers import OracleOCP, OracleUnOCP, OracleUnParNMPC, P, arr, running_sequence, trotting_sequence
    path, flags = native_oracle()
    helpers.ORACLE_PATH_OVERRIDE = path            # the wrappers below load this build
    helpers._oracles.pop(False, None)
    lib = helpers.oracle()
    ncores = physical_cores()                      # SMT siblings add nothing to an FP64-bound loop: "all" = physical cores
    ric = C.c_double()
    set_threads = None
    fz = [0, 0, 0.25 * (-model.total_mass * model.gravity[2])]
    if workload == "anymal_running":
        def make():
            o = OracleOCP(model, cost, cons, T, N, max_num_impulse=nimp)
            running_sequence(o, model, 10)
            o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
            o.init_constraints(0.0)
            return o
        bench, set_threads, nconv = lib.oracle_ocp_bench, lib.oracle_ocp_set_num_threads, 10
    elif workload == "anymal_trotting":
        def make():
            o = OracleOCP(model, cost, cons, T, N, max_num_impulse=nimp + 1)
            trotting_sequence(o, model, nimp)
            o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
            o.init_constraints(0.0)
            return o
        bench, set_threads, nconv = lib.oracle_ocp_bench, lib.oracle_ocp_set_num_threads, 10
    elif workload in ("iiwa14", "iiwa14_task_space"):
        def make():
            o = OracleUnOCP(model, cost, cons, T, N)
            o.set_solution("q", q); o.set_solution("v", v)
            if workload == "iiwa14_task_space":
                from idocp_amd.workloads import task_circle_refs
                o.set_task_refs(task_circle_refs(0.0, T / N, N))
            return o
        bench, set_threads, nconv = lib.oracle_unocp_bench, lib.oracle_unocp_set_num_threads, 50
    elif workload == "iiwa14_unparnmpc":
        def make():
            o = OracleUnParNMPC(model, cost, cons, T, N)
            o.set_solution("q", q); o.set_solution("v", v)
            o.init(0.0)
            return o
        bench, nconv = lib.oracle_unparnmpc_bench, 100
    else:
        def make():
            o = OracleOCP(model, cost, cons, T, N)
            o.set_contact_status([1, 1, 1, 1], pts)
            o.set_solution("q", q); o.set_solution("v", v); o.set_solution("f", fz)
            o.init_constraints(0.0)
            return o
        bench, set_threads, nconv = lib.oracle_ocp_bench, lib.oracle_ocp_set_num_threads, 10
    o = make()
    for _ in range(nconv):                    # converge first (examples/*/…_benchmark.cpp call Convergence before CPUTime)
        o.update(0.0, q, v)
    rows = []
    threads = sorted({1, min(4, ncores), ncores}) if (set_threads is not None and lib.oracle_openmp_enabled()) else [1]
    for nt in threads:
        if set_threads is not None:
            set_threads(o.h, nt)
        t_probe = bench(o.h, 0.0, P(arr(q)), P(arr(v)), 5, C.byref(ric))
        iters = int(max(5, min(20000, target_seconds / max(t_probe / 5, 1e-9))))
        el = bench(o.h, 0.0, P(arr(q)), P(arr(v)), iters, C.byref(ric))
        rows.append({"nthreads": nt, "value": iters / el, "ms_per_update": 1e3 * el / iters, "ms_per_riccati_sweep": 1e3 * ric.value / iters,
                     "updates": iters})
    # THROUGHPUT row: what `value` of the GPU line measures is a batch of independent instances, so the like-for-like CPU number is
    # one independent instance per physical core (each single-threaded, all running at once), not one instance spread over the cores.
    tp = None
    if threaded_ok(lib, set_threads):
        import threading
        k = ncores
        insts = [o] + [make() for _ in range(k - 1)]
        for oi in insts:
            set_threads(oi.h, 1)
        per_inst = max(3, int(rows[0]["value"] * min(target_seconds, 4.0)))      # ~4 s of single-thread work each
        qa, va = arr(q), arr(v)
        gate = threading.Barrier(k + 1)

        def run(oi, fresh):
            if fresh:
                for _ in range(nconv):                # converge first, like the latency rows (in parallel, outside the timed part)
                    oi.update(0.0, q, v)
            gate.wait()
            r = C.c_double()
            bench(oi.h, 0.0, P(qa), P(va), per_inst, C.byref(r))      # (ctypes drops the GIL inside the library)
        ths = [threading.Thread(target=run, args=(oi, i > 0)) for i, oi in enumerate(insts)]
        for th in ths:
            th.start()
        gate.wait()
        t0 = time.perf_counter()
        for th in ths:
            th.join()
        el = time.perf_counter() - t0
        tp = {"instances": k, "threads_per_instance": 1, "updates_per_instance": per_inst, "value": k * per_inst / el,
              "ms_per_update_per_instance": 1e3 * el / per_inst}
    best = max(rows, key=lambda r: r["value"])
    helpers.ORACLE_PATH_OVERRIDE = None
    helpers._oracles.pop(False, None)
    out = {"value": best["value"], "unit": "SQP iterations/s", "cores": best["nthreads"], "kind": "port",
           "sample": "latency rows: updateSolution calls of ONE %s N=%d instance after convergence (reference protocol ocpbenchmarker::CPUTime), ~%.0f s "
                     "per row, OpenMP over the stage loops like the reference (1, 4, all physical cores); throughput row: one independent "
                     "single-threaded instance per physical core, all at once; oracle/ (%s)" % (workload, N, target_seconds, flags),
           "cpu": cpu_model_string(), "host_cores": ncores, "host_threads": os.cpu_count() or 1, "rows": rows, "throughput": tp,
           "ms_per_update": best["ms_per_update"], "ms_per_riccati_sweep": best["ms_per_riccati_sweep"]}
    if tp is not None and tp["value"] > out["value"]:
        # the headline CPU number is the best the host does on this workload: the throughput mode, with the cores it used
        out["value"], out["cores"] = tp["value"], tp["instances"]
    return out


def threaded_ok(lib, set_threads):
    return set_threads is not None and bool(lib.oracle_openmp_enabled())


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of /proc/cpuinfo among the allowed CPUs, capped by the
    cgroup CPU quota (the GPU boxes of this pool show 256 hardware threads but grant 16 CPUs)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cpu, pkg = set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                pkg = int(line.split(":")[1])
            elif line.startswith("core id") and cpu in allowed:
                cores.add((pkg, int(line.split(":")[1])))
    except (OSError, ValueError):
        pass
    n = len(cores) or len(allowed) or 1
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) bounds what those cores can deliver
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = max(1, min(n, quota // period))
        except (OSError, ValueError):
            pass
    return n


def init_distributed(backend, local_rank):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    backend "nccl" is RCCL on ROCm; the CPU test of this scaffolding uses "gloo"."""
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def run_timed(step, sync, steps, warmup, dist, device, events=None):
    """The timing protocol of the contract: `warmup` untimed steps, then EXACTLY `steps` steps
    bracketed by barrier + synchronize on both sides; returns the MAX elapsed seconds over ranks.
    The replicas exchange nothing else (the Riccati path does not shard)."""
    for _ in range(warmup):
        step(None)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k] if events is not None else None)
    sync()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    return el


def whole_job_value(world, batch_per_rank, steps, elapsed):
    """SQP iterations per second summed over all ranks (weak scaling: every rank owns `batch_per_rank`
    independent OCP instances)."""
    return world * batch_per_rank * steps / elapsed


def latency_mode(lib, hip, build, q0, v0, iters=40):
    """The "ms / Riccati sweep" half of the metric: ONE OCP instance (batch 1, what one reference solver object is), the same
    problem as the throughput run.  ms per SQP iteration with plain launches and with the hipGraph replay
    (idocp_ocp_update_solution_graph), and the backward + forward Riccati sweep over the chain from HIP events."""
    from idocp_amd import capi
    sv = build(1)
    q1, v1 = np.ascontiguousarray(q0[:1]), np.ascontiguousarray(v0[:1])
    d_q, d_v = C.c_void_p(), C.c_void_p()
    capi.check(lib.idocp_device_alloc(C.byref(d_q), q1.nbytes))
    capi.check(lib.idocp_device_alloc(C.byref(d_v), v1.nbytes))
    capi.check(lib.idocp_device_upload(d_q, q1.ctypes.data, q1.nbytes))
    capi.check(lib.idocp_device_upload(d_v, v1.ctypes.data, v1.nbytes))
    stream = lib.idocp_ocp_stream(sv.h)
    out = {"batch": 1}
    for name, fn in (("ms_per_iteration", lib.idocp_ocp_update_solution_device), ("ms_per_iteration_hipgraph", lib.idocp_ocp_update_solution_graph)):
        for _ in range(5):
            capi.check(fn(sv.h, 0.0, d_q, d_v), name)
        capi.check(lib.idocp_ocp_synchronize(sv.h))
        t0 = time.perf_counter()
        for _ in range(iters):
            capi.check(fn(sv.h, 0.0, d_q, d_v), name)
        capi.check(lib.idocp_ocp_synchronize(sv.h))
        out[name] = 1e3 * (time.perf_counter() - t0) / iters
    ev = [hip.event() for _ in range(3)]
    acc = [0.0, 0.0]
    for _ in range(iters):
        for kid in range(len(KERNELS_OCP)):              # (ids 4, 5 launch nothing when the forward sweep expands as it walks)
            if kid in (2, 3):
                hip.record(ev[kid - 2], stream)
            capi.check(lib.idocp_ocp_launch_kernel(sv.h, kid, d_q, d_v), "kernel %d" % kid)
            if kid == 3:
                hip.record(ev[2], stream)
        capi.check(lib.idocp_ocp_synchronize(sv.h))
        acc[0] += hip.elapsed_ms(ev[0], ev[1])
        acc[1] += hip.elapsed_ms(ev[1], ev[2])
    out["ms_riccati_backward"] = acc[0] / iters
    out["ms_riccati_forward"] = acc[1] / iters
    out["ms_per_riccati_sweep"] = (acc[0] + acc[1]) / iters
    kkt = sv.kkt_error(0.0, q1, v1)
    assert np.isfinite(kkt).all()
    return out


def spawn_ranks(n, argv, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) with
    torch.distributed.run and hand back their exit code.  Called BEFORE this process imports torch or makes any
    HIP call -- a process that has initialised the GPU must never be replaced or forked into ranks.  The ranks
    inherit stdout, so rank 0's JSON line is this command's output.
    A first contact between ranks that never completes (an RCCL rendezvous, a halo whose peer is missing) must not eat the caller's
    whole time budget: after `timeout_s` (--timeout / IDOCP_BENCH_TIMEOUT, default 1200 s) the launcher's process group is ended
    -- these are exactly the processes started here, in a session of their own -- and the command exits with 124 and the tail
    of what the ranks wrote to stderr."""
    import signal
    import socket
    import subprocess
    import tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get("IDOCP_BENCH_TIMEOUT", "1200"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    def descendants(pid):
        """every process below `pid` in the parent tree (/proc): torchrun gives its workers sessions of their own, so the ranks are NOT in the
        launcher's process group -- they are found by parentage, before anything is signalled (an orphan is re-parented and lost)"""
        par, born = {}, {}
        try:
            for d in os.listdir("/proc"):
                if d.isdigit():
                    try:
                        st = open("/proc/%s/stat" % d).read()
                        f = st[st.rindex(")") + 2:].split()
                        par[int(d)], born[int(d)] = int(f[1]), f[19]          # parent pid, start time: (pid, start time) names a process for good
                    except (OSError, ValueError, IndexError):
                        pass
        except OSError:
            return []
        out, frontier = [], [pid]
        while frontier:
            nxt = [p for p, pp in par.items() if pp in frontier]
            out += [(p, born[p]) for p in nxt]
            frontier = nxt
        return out

    def still(p, start):
        try:
            st = open("/proc/%d/stat" % p).read()
            return st[st.rindex(")") + 2:].split()[19] == start
        except (OSError, ValueError, IndexError):
            return False

    def end_group(proc):
        """SIGTERM to the launcher's process group (torchrun forwards it to its workers), then SIGKILL to the group AND to every process that
        descended from it -- exactly the processes started here -- so that a rank stuck inside a driver call cannot outlive the command"""
        tree = descendants(proc.pid)
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        for p, start in tree:                            # survivors of the polite round (a rank that ignores SIGTERM); never a recycled pid
            if still(p, start):
                try:
                    os.kill(p, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass

    with tempfile.TemporaryFile(mode="w+") as err:
        proc = subprocess.Popen(cmd, stderr=err, start_new_session=True)
        # The ranks live in a session of their own, so a signal aimed at THIS process (a driver's SIGTERM, Ctrl-C) does not reach them:
        # forward it -- SIGTERM / SIGINT / SIGHUP end the group before this process exits with 128 + signo -- and whatever else
        # unwinds the wait (KeyboardInterrupt, an exception) ends the group in the `finally`.  No rank outlives its launcher.
        interrupted = []

        def forward(signo, _frame):
            interrupted.append(signo)
            raise KeyboardInterrupt

        old = {}
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            try:
                old[sg] = signal.signal(sg, forward)
            except (ValueError, OSError):      # (not the main thread: the caller owns the signals)
                pass
        timed_out = False
        try:
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                timed_out = True
                rc = 124
        except KeyboardInterrupt:
            rc = 128 + (interrupted[0] if interrupted else signal.SIGINT)
        finally:
            if proc.poll() is None:
                end_group(proc)
            for sg, h in old.items():
                signal.signal(sg, h)
        err.seek(0)
        text = err.read()
    if interrupted:
        sys.stderr.write("bench.py: launcher received signal %d -- ended the %d ranks.\n" % (interrupted[0], n))
        return rc
    if timed_out:
        sys.stderr.write("bench.py: the %d ranks did not finish within %.0f s -- ended them.  Last stderr of the ranks:\n%s\n" % (n, timeout_s, text[-4000:]))
    elif text:
        sys.stderr.write(text)
    return rc


def run_stub(args, rank, world, dist):
    """Test scaffolding (IDOCP_BENCH_STUB=1, tests/test_bench_dist.py): the launcher, the rendezvous, the timing protocol
    and the JSON line of the replicas path with a stand-in step that needs no GPU (backend from IDOCP_BENCH_BACKEND)."""
    B = args.batch or 8
    a = np.ones((64, 64))

    def step(_events):
        np.dot(a, a)
        time.sleep(0.002 * (1 + rank))

    el = run_timed(step, lambda: None, args.steps, args.warmup, dist, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": whole_job_value(world, B, args.steps, el), "unit": "SQP iterations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "scaling": "weak",
                          "config": {"workload": "stub (no GPU)", "batch_per_gpu": B}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def exchange_unique_id(lib, dist, rank, world):
    """The 128-byte RCCL unique id: made on rank 0 by the library (idocp_comm_get_unique_id), handed to the other ranks over the
    gloo group that otherwise only carries the barrier / max-reduction of the timing.  Returns the ctypes buffer every rank passes to
    idocp_comm_init_rank.  (tests/test_bench_dist.py runs this on two gloo processes.)"""
    nbytes = 128                                     # IDOCP_COMM_ID_BYTES
    raw = (C.c_char * nbytes)()
    if rank == 0:
        rc = lib.idocp_comm_get_unique_id(raw)
        if rc != 0:
            raise RuntimeError("idocp_comm_get_unique_id failed (%d)" % rc)
    if world > 1:
        import torch
        idbuf = torch.tensor(list(raw.raw), dtype=torch.uint8) if rank == 0 else torch.zeros(nbytes, dtype=torch.uint8)
        dist.broadcast(idbuf, src=0)
        raw = (C.c_char * nbytes).from_buffer_copy(bytes(idbuf.tolist()))
    return raw


class _StubDistLib:
    """IDOCP_BENCH_STUB=1 (no GPU): stands in for the idocp_comm_* / idocp_parnmpc_dist_* entry points so that the launcher, the
    unique-id hand-off over gloo, the shard arithmetic and the JSON line of `--workload anymal_parnmpc --gpus N` run on CPU."""

    def __init__(self):
        self.id_seen = None
        self.updates = 0

    def idocp_comm_get_unique_id(self, raw):
        pattern = bytes((37 * i + 11) % 251 for i in range(128))
        C.memmove(raw, pattern, 128)
        return 0

    def idocp_comm_init_rank(self, raw, rank, world, device, out):
        self.id_seen = bytes(raw.raw)
        assert self.id_seen == bytes((37 * i + 11) % 251 for i in range(128)), "rank %d received a corrupted unique id" % rank
        return 0

    def idocp_parnmpc_dist_update_solution(self, h, t):
        self.updates += 1
        time.sleep(0.002)
        if os.environ.get("IDOCP_BENCH_STUB_HANG") and self.updates == 2:
            time.sleep(3600)                        # tests/test_bench_dist.py: a first contact that never completes
        return 0


def run_parnmpc_stub(args, rank, local_rank, world, dist):
    """The anymal_parnmpc bench path with the library calls replaced by _StubDistLib (CPU test of the multi-rank plumbing)."""
    N = args.horizon if args.horizon != 100 else 256
    B = args.batch or 256
    if N % world:
        raise SystemExit("--horizon must be divisible by the number of GPUs")
    lib = _StubDistLib()
    raw = exchange_unique_id(lib, dist, rank, world)
    assert lib.idocp_comm_init_rank(raw, rank, world, local_rank, None) == 0

    def step(_events):
        assert lib.idocp_parnmpc_dist_update_solution(None, 0.0) == 0

    el = run_timed(step, lambda: None, args.steps, args.warmup, dist, "cpu")
    assert lib.updates == args.steps + args.warmup
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": B * args.steps / el, "unit": "SQP iterations/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "scaling": "strong",
                          "config": {"workload": "stub (no GPU): anymal_parnmpc plumbing", "horizon": N, "batch_per_gpu": B,
                                     "parallelism": "horizon shards x%d" % world, "stages_per_rank": N // world}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# what the horizon shards of configs[3] can buy.  The stage-parallel kernels divide by G; the two serial correction
# sweeps are dependency chains over the N stages of an instance and cost N stage-times on any number of GPUs, plus 2 (G - 1) hops
# (pack + ncclSend / ncclRecv + unpack of a 74 kB halo, ~15 us over xGMI), the boundary exchange and two all-reduces (~0.1 ms).
# On one GPU the two parts are this run's own HIP-event times; with G > 1 ranks they are the one-GPU figures of the committed
# round-5 profile (profiles/r05_anymal_parnmpc_kernel_trace.txt), labelled as such.
PARNMPC_MODEL_1GPU = {"anymal_parnmpc": {"stage_parallel_ms": 5.25, "sweeps_ms": 0.95}, "anymal_parnmpc_trotting": {"stage_parallel_ms": 6.8, "sweeps_ms": 1.1}}


def parnmpc_model_ms(workload, world, ker=None):
    if ker:
        serial = ker.get("parnmpc_backward_serial", 0.0) + ker.get("parnmpc_forward_serial", 0.0)
        par, src = sum(ker.values()) - serial, "this run's kernel times"
    else:
        m = PARNMPC_MODEL_1GPU[workload]
        par, serial, src = m["stage_parallel_ms"], m["sweeps_ms"], "one-GPU kernel times of the committed profile (bench.PARNMPC_MODEL_1GPU)"
    model = {}
    for g in (1, 2, 4, 8):
        comm = 0.0 if g == 1 else 2 * (g - 1) * 0.015 + 0.1
        model[str(g)] = par / g + serial + comm
    return {"formula": "stage_parallel / G + sweeps + 2 (G - 1) * 0.015 ms + 0.1 ms (DESIGN.md section 5)", "stage_parallel_ms": par, "sweeps_ms": serial,
            "source": src, "ms_per_step_by_gpus": model, "this_run": model[str(world)] if str(world) in model else None}


def run_parnmpc_cxx(args, rank, local_rank, world, dist):
    """BASELINE.json configs[3]: ANYmal ParNMPCSolver, N = 256 (T = 12.8, dt = 0.05), the stages of every instance sharded over the
    ranks and driven by the library's C++ multi-GPU driver (idocp_amd/csrc/parnmpc_dist.hip: RCCL point-to-point halos with the two
    neighbours + all-reduce of the step sizes, everything enqueued on the shard's stream).  Strong scaling: the batch of instances is
    the same whatever the number of GPUs.
    --workload anymal_parnmpc: 4 point contacts on every stage; the iterate is WARM-STARTED from the converged Riccati solution of the
    same OCP (workloads.warm_start_parnmpc; the MPC use of the solver): from the reference's cold start the forward correction sweep
    amplifies by 1.15 per stage and a 256-stage horizon is numerically meaningless.
    --workload anymal_parnmpc_trotting: configs[3] as BASELINE.json words it -- the TROTTING contact sequence of
    examples/anymal/anymal_trotting.cpp over the whole horizon (1 lift + 24 touch-down events a quarter of a time step off the grid:
    305 stages in the chain, sharded by idocp_parnmpc_create_hybrid_shard).  The iterate MOVES: it starts from a converged ParNMPC
    solution of this very problem, found by continuation in the step length from the trot in place (workloads.
    trotting_parnmpc_by_continuation, untimed set-up on a batch-1 solver; every rank runs it for itself and loads its slice of the chain),
    the instances' initial states are then spread, and the timed iterations re-converge from there.  (Rounds 1-3 timed this workload with
    the iterate frozen at a standing guess, through a Python driver.)"""
    from idocp_amd import capi, workloads
    from idocp_amd.workloads import (ANYMAL_Q_STANDING, HipOCP, P, ParNMPCShardHandle, anymal_contact_points, anymal_model, anymal_problem, arr,
                                     trotting_parnmpc_by_continuation, trotting_sequence)
    os.environ.pop("NCCL_DEBUG", None)              # (the pool exports NCCL_DEBUG=VERSION, and RCCL logs to STDOUT: its banner -- at WARN its
                                                    #  "could not read node" notes -- would land inside the one JSON line of the contract)
    lib = capi.lib()
    hip = Hip()
    hip.rt.hipSetDevice(local_rank)
    trot = args.workload == "anymal_parnmpc_trotting"
    N = args.horizon if args.horizon != 100 else 256
    T = 0.05 * N
    B = args.batch or 256
    if N % world:
        raise SystemExit("--horizon must be divisible by the number of GPUs")
    Nl = N // world
    model = anymal_model()
    cost, cons = anymal_problem(model, trotting_ref=True)
    pts = anymal_contact_points(model)
    nq, nv = model.nq, model.nv
    rng = np.random.default_rng(20250)
    q0 = np.tile(ANYMAL_Q_STANDING, (B, 1))
    q0[:, 7:] += (0.002 if trot else 0.02) * rng.uniform(-1, 1, (B, 12))
    q0 = np.ascontiguousarray(q0)
    v0 = np.zeros((B, nv))
    fz = [0, 0, 0.25 * (-model.total_mass * model.gravity[2])]
    n_events = (int((T - 0.5125) / 0.5) + 1) if trot else 0
    setup = {}
    if trot:
        # ---- the converged moving trot (batch 1, this GPU), then this rank's shard ----
        t0 = time.perf_counter()
        clog = []
        cost, pn, Mp = trotting_parnmpc_by_continuation(model, T, N, device=local_rank, log=clog)
        setup = {"continuation_steps": len(clog), "continuation_failed_steps": sum(1 for c in clog if not c["ok"]),
                 "continuation_seconds": time.perf_counter() - t0, "kkt_of_the_start": clog[-1]["kkt"]}
        gchain = pn.chain(0.0)[:-1]
        gsol = {f: pn.get_chain(f, Mp + 1)[:Mp] for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu")}
        gaux = np.zeros((Mp + 1, 4 * nv * nv))
        capi.check(lib.idocp_parnmpc_get_aux_mat_chain(pn.h, 0, P(gaux)), "get_aux_mat_chain")
        del pn
        shard = ParNMPCShardHandle(model, cost, cons, T, N, rank, world, B, local_rank, max_num_impulse=n_events)
        trotting_sequence(shard, model, n_events - 1, t_start=0.5125)
    else:
        # ---- warm start: the OCP of the same problem (nominal initial state), converged on this GPU ----
        src = HipOCP(model, cost, cons, T, N, batch=1, device=local_rank)
        src.set_contact_status([1, 1, 1, 1], pts)
        src.set_solution("q", ANYMAL_Q_STANDING)
        src.set_solution("v", np.zeros(nv))
        src.set_solution("f", fz)
        src.init_constraints(0.0)
        qn, vn = ANYMAL_Q_STANDING.copy(), np.zeros(nv)
        for _ in range(40):
            assert src.update(0.0, qn, vn) == 0
            if src.kkt_error(0.0, qn, vn)[0] < 1e-9:
                break
        shard = ParNMPCShardHandle(model, cost, cons, T, N, rank, world, B, local_rank)
        shard.set_contact_status([1, 1, 1, 1], pts)
    for name, val in (("q", ANYMAL_Q_STANDING), ("v", np.zeros(nv)), ("f", fz)):
        shard.set_solution(name, val)
    # the real RCCL communicator at every world size, one GPU included: `--gpus 1` runs ncclCommInitRank and the driver's stream-ordered
    # transport exactly like a rank of the 8-GPU job (no in-process substitute)
    comm = C.c_void_p()
    raw = exchange_unique_id(lib, dist, rank, world)
    capi.check(lib.idocp_comm_init_rank(raw, rank, world, local_rank, C.byref(comm)), "comm_init_rank")
    capi.check(lib.idocp_parnmpc_dist_attach(shard.h, comm), "dist_attach")
    # what RCCL itself says it connected, and -- BEFORE anything is timed -- every halo kind across the REAL neighbours in the driver's own
    # grouping, the all-reduces and the broadcast against known patterns (idocp_parnmpc_dist_transport_selftest; world 1: loop-back)
    lib.idocp_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    lib.idocp_parnmpc_dist_transport_selftest.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    nr, ur, ver, tr = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    capi.check(lib.idocp_comm_info(comm, C.byref(nr), C.byref(ur), C.byref(ver), C.byref(tr)), "comm_info")
    if nr.value != world or ur.value != rank:
        raise SystemExit("bench.py: RCCL reports rank %d of %d, the launcher started rank %d of %d" % (ur.value, nr.value, rank, world))
    dev = C.c_double(-1.0)
    rc_self = lib.idocp_parnmpc_dist_transport_selftest(shard.h, C.byref(dev))
    worst = np.array([dev.value if rc_self == 0 else np.inf])
    if dist is not None:
        import torch
        tw = torch.from_numpy(worst)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    if not (worst[0] <= 1e-9):
        raise SystemExit("bench.py: transport self-test FAILED on rank %d of %d before the timed region: rc %d, largest deviation over the ranks %r (%s)"
                         % (rank, world, rc_self, float(worst[0]), lib.idocp_last_error().decode() if rc_self else "halo / collective contents differ from the pattern"))
    rccl_info = {"rccl_nranks": nr.value, "rccl_version": ver.value, "transport": "rccl" if tr.value else "in-process",
                 "selftest_max_abs_diff": float(worst[0])}
    if rank == 0:
        capi.check(lib.idocp_parnmpc_dist_set_initial_state(shard.h, P(q0), P(v0), nq, nv))
    capi.check(lib.idocp_parnmpc_dist_init_backward_correction(shard.h, 0.0), "dist_init_backward_correction")
    if trot:
        # this rank's slice of the chain: the nodes of the shard's own chain (without its placeholder), found in the whole chain by kind / index
        where = {(c["kind"], c["index"]): p for p, c in enumerate(gchain)}
        lchain = shard.chain(0.0)[:-1]
        ids = [where[(c["kind"], c["index"])] for c in lchain]
        for f, x in gsol.items():
            vals = arr(x[ids])
            capi.check(lib.idocp_ocp_set_solution_chain(shard.h, f.encode(), len(ids), P(vals)), "set_solution_chain " + f)
        capi.check(lib.idocp_parnmpc_set_aux_mat_chain(shard.h, len(ids), P(arr(gaux[ids]))), "set_aux_mat_chain")
    else:
        class _Slice:                                   # workloads.warm_start_parnmpc target: this rank's stages of the horizon
            def set_stage_values(self, name, values):
                vals = arr(values[rank * Nl:(rank + 1) * Nl])
                capi.check(lib.idocp_ocp_set_solution_stages(shard.h, name.encode(), Nl, P(vals)), "set_solution_stages")

            def set_aux_mats(self, mats):
                cm = arr(np.asarray(mats[rank * Nl:(rank + 1) * Nl]).transpose(0, 2, 1))
                capi.check(lib.idocp_parnmpc_set_aux_mat(shard.h, Nl, P(cm)), "set_aux_mat")
        workloads.warm_start_parnmpc(src, [_Slice()], N)
        del src
    capi.check(lib.idocp_ocp_init_constraints(shard.h, 0.0))

    def step(_events):
        capi.check(lib.idocp_parnmpc_dist_update_solution(shard.h, 0.0), "dist_update_solution")

    def sync():
        capi.check(lib.idocp_ocp_synchronize(shard.h))
        hip.rt.hipDeviceSynchronize()

    el = run_timed(step, sync, args.steps, args.warmup, dist, "cpu")
    kkt = np.zeros(B)
    capi.check(lib.idocp_parnmpc_dist_kkt_error(shard.h, 0.0, P(kkt)), "dist_kkt_error")
    assert np.isfinite(kkt).all(), "non-finite KKT error after the timed region"
    # per-kernel durations (HIP events on the shard's stream, one extra iteration phase by phase; single GPU only: with several
    # ranks the phases interleave with the halo traffic and are timed by the iteration as a whole)
    ker = {}
    if world == 1:
        stream = lib.idocp_ocp_stream(shard.h)
        PH = ["ocp_rnea", "ocp_condense", "parnmpc_kkt_inverse", "parnmpc_backward_serial", "parnmpc_backward_parallel",
              "parnmpc_forward_serial", "parnmpc_forward_parallel", "ocp_expand_primal", "ocp_reduce_steps", "ocp_expand_dual_integrate"]
        reps = 3
        for _ in range(reps):
            capi.check(lib.idocp_parnmpc_discretize(shard.h, 0.0))
            ev = [hip.event() for _ in range(len(PH) + 1)]
            for k in range(len(PH)):
                hip.record(ev[k], stream)
                capi.check(lib.idocp_parnmpc_launch_phase(shard.h, k, shard.d_q, shard.d_v), PH[k])
            hip.record(ev[len(PH)], stream)
            capi.check(lib.idocp_ocp_synchronize(shard.h))
            for k in range(len(PH)):
                ker[PH[k]] = ker.get(PH[k], 0.0) + hip.elapsed_ms(ev[k], ev[k + 1]) / reps
    if rank == 0:
        ms_step = 1e3 * el / args.steps
        out = {
            "metric": "SQP iterations/sec (whole node)", "value": B * args.steps / el, "unit": "SQP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("ANYmal ParNMPCSolver N=%d T=%.2f FP64, %s (BASELINE.json configs[3]); batch=%d OCP instances, the %d stages "
                                    "of every instance sharded over %d GPU(s), C++ driver with RCCL halo exchange (idocp_parnmpc_dist_*)"
                                    % (N, T, ("TROTTING contact sequence (1 lift + %d impulse events: %d stages in the chain), moving iterate started from "
                                              "the converged ParNMPC solution of this problem (continuation in the step length)" % (n_events - 1, N + 2 * (n_events - 1) + 1))
                                       if trot else "4 point contacts on every stage, warm-started from the converged Riccati solution of the same OCP", B, N, world)),
                       "horizon": N, "batch_per_gpu": B, "parallelism": "horizon shards x%d" % world,
                       "stages_per_rank": Nl, "setup": setup, "rccl": rccl_info,
                       "model_ms": parnmpc_model_ms(args.workload, world, ker if world == 1 else None),
                       "kernel_ms": ker, "max_kkt_error_after": float(kkt.max())},
        }
        if ker:
            dom = max(ker, key=ker.get)
            alg_bytes = A_STAGE["anymal_parnmpc"] * B * (Nl + (2 * (n_events - 1) + 1 if trot else 0))
            achieved = alg_bytes / (ker[dom] * 1e-3) / 1e9
            # (the bench's phase "parnmpc_kkt_inverse" is the launch of parnmpc_kkt_inverse_wave_kernel unless IDOCP_K9_WAVE=0)
            pmc_name = "parnmpc_kkt_inverse_wave" if dom == "parnmpc_kkt_inverse" and os.environ.get("IDOCP_K9_WAVE", "1") != "0" else dom
            traffic, traffic_source = quoted_traffic(args.workload, B, N, pmc_name)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg_bytes,
                               "avg_launch_ms": ker[dom],
                               "whole_step_frac": A_STAGE["anymal_parnmpc"] * B * N / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world}
            # FP64 roofline of the dominant kernel (exact FLOPs of the restatement's formulation, tests/golden/oracle_flops.json) and its issue counters
            frec = None if trot else oracle_flops_record(args.workload, N)
            if frec is not None:
                kf = kernel_flops(frec, dom)
                if kf:
                    tfl = B * kf / (ker[dom] * 1e-3) / 1e12
                    out["roofline"]["flops"] = {"per_launch": B * kf, "achieved_tflops": tfl, "peak": FP64_PEAK_TFLOPS, "frac": tfl / FP64_PEAK_TFLOPS,
                                                "whole_step_tflops": B * frec["flop_per_iteration"] / (ms_step * 1e-3) / 1e12,
                                                "whole_step_frac": B * frec["flop_per_iteration"] / (ms_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                "source": "tests/golden/oracle_flops.json (exact operation counts of the CPU restatement, FMA = 2)"}
                    sq, sq_source = quoted_sq(args.workload, B, N, pmc_name)
                    if sq is not None:
                        out["roofline"]["flops"]["lane_utilisation"] = (B * kf / 2) / max(1.0, sq["valu_lane_slots_per_launch"])
                        out["roofline"]["mfma_util"] = sq.get("mfma_util")
                    out["roofline"]["sq_source"] = sq_source
        if not args.no_cpu_baseline and world == 1 and not trot:
            from helpers import OracleParNMPC      # the CPU restatement: checker / baseline only
            o = OracleParNMPC(model, cost, cons, T, N)
            o.set_contact_status([1, 1, 1, 1], pts)
            o.set_solution("q", ANYMAL_Q_STANDING)
            o.set_solution("v", np.zeros(nv))
            o.set_solution("f", fz)
            o.init(0.0)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 12.0:
                o.update(0.0, q0[0], v0[0])
                n += 1
            elc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n / elc, "unit": "SQP iterations/s", "cores": 1, "kind": "port", "cpu": cpu_model_string(),
                                   "sample": "%d updateSolution calls of one ParNMPC N=%d instance, single thread (oracle/, -O3)" % (n, N),
                                   "ms_per_update": 1e3 * elc / n}
        print(json.dumps(out), flush=True)
    capi.check(lib.idocp_parnmpc_dist_detach(shard.h))
    lib.idocp_comm_destroy(comm)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["anymal", "anymal_trotting", "anymal_running", "anymal_parnmpc", "anymal_parnmpc_trotting", "iiwa14", "iiwa14_task_space", "iiwa14_unparnmpc"], default="anymal_trotting",
                    help="anymal_trotting = BASELINE.json configs[2] (trotting contact sequence); anymal = its uniform 4-contact variant "
                         "(SURVEY 8d roofline case); iiwa14 = configs[1]; anymal_parnmpc = configs[3] (ParNMPC, N=256, the horizon "
                         "sharded over the ranks, strong scaling)")
    ap.add_argument("--batch", type=int, default=0, help="independent OCP instances per GPU (0 = workload default)")
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--friction-cone", choices=["linearized", "nonlinear"], default="linearized",
                    help="ANYmal workloads: LinearizedFrictionCone (the trotting / running examples) or FrictionCone (examples/anymal/ocp_benchmark.cpp:76)")
    ap.add_argument("--timeout", type=float, default=None, help="--gpus N > 1 started without a launcher: seconds after which the ranks are ended "
                                                              "and the command exits with 124 (default: IDOCP_BENCH_TIMEOUT or 1200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement (config.latency)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N`: this process has not touched the GPU (numpy / ctypes only so far); it becomes the launcher
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    backend = os.environ.get("IDOCP_BENCH_BACKEND", "nccl")
    if args.workload in ("anymal_parnmpc", "anymal_parnmpc_trotting"):
        # the data path is the library's own RCCL communicator (idocp_parnmpc_dist_*); torch.distributed only carries the
        # rendezvous of its id and the barrier / max-reduction of the timing -> gloo, so that one RCCL instance owns the GPUs
        dist = init_distributed("gloo", local_rank) if world > 1 else None
        if os.environ.get("IDOCP_BENCH_STUB"):
            return run_parnmpc_stub(args, rank, local_rank, world, dist)
        return run_parnmpc_cxx(args, rank, local_rank, world, dist)
    dist = None
    if world > 1 or os.environ.get("IDOCP_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL scaffolding on one GPU
        dist = init_distributed(backend, local_rank)
    if os.environ.get("IDOCP_BENCH_STUB"):
        return run_stub(args, rank, world, dist)

    from idocp_amd import capi
    from idocp_amd.workloads import (ANYMAL_Q_STANDING, HipOCP, HipUnOCP, HipUnParNMPC, anymal_contact_points, anymal_model, anymal_problem,
                                     iiwa14_model, trotting_sequence, unocp_problem)
    lib = capi.lib()                       # fails loudly if the HIP extension is missing
    hip = Hip()
    hip.rt.hipSetDevice(local_rank)

    N = args.horizon
    T = 0.05 * N
    rng = np.random.default_rng(20240 + rank)
    pts = None
    nimp = 0
    KIDS = None          # kernel ids behind the names of KERNELS where they are not 0, 1, 2, ...
    if args.workload == "anymal_trotting":
        # ANYmal OCPSolver, N=100, T=5.05, trotting schedule of examples/anymal/anymal_trotting.cpp:144-177 with
        # 9 impulse phases (1 lift + 9 impulse events: 120 stages in the chain), switching constraints, impulse stages
        nimp = 9
        T = 0.5 + nimp * 0.5 + 0.05
        B = args.batch or 1024
        model = anymal_model()
        cost, cons = anymal_problem(model, trotting_ref=True, cone=args.friction_cone)
        nq, nv = model.nq, model.nv
        # the batch of SURVEY 8(d) C3: instance b of the node starts from q_standing with base-xy and joints moved by 0.02 U(-1, 1) drawn from a
        # generator seeded 20250 + b (numpy's generator here, std::mt19937_64 there), v = 0; the quaternion stays the identity
        q0 = np.tile(ANYMAL_Q_STANDING, (B, 1))
        for bi in range(B):
            r_b = np.random.default_rng(20250 + rank * B + bi)
            q0[bi, 0:2] += 0.02 * r_b.uniform(-1, 1, 2)
            q0[bi, 7:] += 0.02 * r_b.uniform(-1, 1, 12)
        v0 = np.zeros((B, nv))
        def build(batch):
            sv = HipOCP(model, cost, cons, T, N, batch=batch, device=local_rank, max_num_impulse=nimp + 1)
            trotting_sequence(sv, model, nimp)
            sv.set_solution("q", ANYMAL_Q_STANDING)
            sv.set_solution("v", np.zeros(nv))
            sv.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
            sv.init_constraints(0.0)
            return sv
        solver = build(B)
        Mc = len(solver.chain(0.0))
        KERNELS, KIDS, units, riccati_ids = ocp_kernel_lists(lib, solver.h, B, Mc)
        launch, sync_fn, stream = lib.idocp_ocp_launch_kernel, lib.idocp_ocp_synchronize, lib.idocp_ocp_stream(solver.h)
        desc = ("ANYmal OCPSolver N=%d T=%.2f FP64, trotting contact sequence (1 lift + %d impulse events, %d stages incl. impulse / "
                "aux / lift stages, switching constraints; BASELINE.json configs[2]); " % (N, T, nimp, Mc))
        assert solver.update(0.0, q0, v0) == 0
    elif args.workload == "anymal_running":
        # BASELINE.json configs[4] in FP64: ANYmal OCPSolver on the running gait of examples/anymal/anymal_running.cpp:29-231
        # (40 discrete events: 26 touch-downs, 14 lift-offs, flight phases without any contact), N = 200, T = 7
        from idocp_amd.workloads import ANYMAL_Q_RUNNING_START, running_problem, running_sequence
        N = args.horizon if args.horizon != 100 else 200
        T = 7.0 * N / 240                           # the example's own time step (examples/anymal/anymal_running.cpp: T = 7, N = 240)
        nimp = 26
        B = args.batch or 1024      # (512 until the end of round 4: the backward Riccati sweep runs one wavefront per instance, half of the SIMDs idle)
        model = anymal_model()
        cost, cons = running_problem(model, 10)
        nq, nv = model.nq, model.nv
        q0 = np.tile(ANYMAL_Q_RUNNING_START, (B, 1))
        v0 = np.zeros((B, nv))
        def build(batch):
            sv = HipOCP(model, cost, cons, T, N, batch=batch, device=local_rank, max_num_impulse=nimp)
            running_sequence(sv, model, 10)
            sv.set_solution("q", ANYMAL_Q_RUNNING_START)
            sv.set_solution("v", np.zeros(nv))
            sv.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
            sv.init_constraints(0.0)
            return sv
        solver = build(B)
        Mc = len(solver.chain(0.0))
        KERNELS, KIDS, units, riccati_ids = ocp_kernel_lists(lib, solver.h, B, Mc)
        launch, sync_fn, stream = lib.idocp_ocp_launch_kernel, lib.idocp_ocp_synchronize, lib.idocp_ocp_stream(solver.h)
        desc = ("ANYmal OCPSolver N=%d T=%.2f FP64, running contact sequence of examples/anymal/anymal_running.cpp (26 impulse + 14 lift "
                "events, flight phases, %d stages in the chain; BASELINE.json configs[4] in FP64); " % (N, T, Mc))
        assert solver.update(0.0, q0, v0) == 0
    elif args.workload == "anymal":
        # BASELINE.json configs[2] / metric config: ANYmal OCPSolver, N=100, T=5 (dt=0.05), 4 point contacts active on
        # every stage (the uniform-contact variant of SURVEY 8d C3: trotting cost + linearized friction cone), FP64
        B = args.batch or 1024
        model = anymal_model()
        cost, cons = anymal_problem(model, trotting_ref=True, cone=args.friction_cone)
        pts = anymal_contact_points(model)
        nq, nv = model.nq, model.nv
        q0 = np.tile(ANYMAL_Q_STANDING, (B, 1))
        q0[:, 0:2] += 0.02 * rng.uniform(-1, 1, (B, 2))
        q0[:, 7:] += 0.02 * rng.uniform(-1, 1, (B, 12))
        q0 = np.ascontiguousarray(q0)
        v0 = np.zeros((B, nv))
        def build(batch):
            sv = HipOCP(model, cost, cons, T, N, batch=batch, device=local_rank)
            sv.set_contact_status([1, 1, 1, 1], pts)
            sv.set_solution_batch("q", np.ascontiguousarray(q0[:batch]))
            sv.set_solution("v", v0[0])
            sv.set_solution("f", [0, 0, 0.25 * (-model.total_mass * model.gravity[2])])
            sv.init_constraints(0.0)
            return sv
        solver = build(B)
        KERNELS, KIDS, units, riccati_ids = ocp_kernel_lists(lib, solver.h, B, N + 1)
        launch, sync_fn, stream = lib.idocp_ocp_launch_kernel, lib.idocp_ocp_synchronize, lib.idocp_ocp_stream(solver.h)
        desc = ("ANYmal OCPSolver N=%d T=%.2f FP64, 4 point contacts active on every stage (BASELINE.json configs[2], "
                "uniform-contact variant), trotting cost + joint limits + linearized friction cone; " % (N, T))
        assert solver.update(0.0, q0, v0) == 0            # one full update through the host entry (uploads the stage references)
    elif args.workload == "iiwa14_unparnmpc":
        # iiwa14 UnParNMPCSolver (examples/iiwa14/unparnmpc_benchmark.cpp at the horizon of configs[1]), FP64
        B = args.batch or 8192
        model = iiwa14_model()
        cost, cons = unocp_problem(model)
        nq, nv = model.nq, model.nv
        q0 = np.ascontiguousarray(2.0 + 0.1 * rng.uniform(-1, 1, (B, nv)))
        v0 = np.zeros((B, nv))
        q0 = np.ascontiguousarray(1.0 + 0.1 * rng.uniform(-1, 1, (B, nv)))     # inside the joint limits: the stage-wise method needs a feasible start
        solver = HipUnParNMPC(model, cost, cons, T, N, batch=B, device=local_rank)
        solver.set_solution_batch("q", q0)
        solver.set_solution("v", v0[0])
        solver.init(0.0)
        KERNELS = KERNELS_UNP
        launch, sync_fn, stream = lib.idocp_unparnmpc_launch_phase, lib.idocp_unocp_synchronize, lib.idocp_unocp_stream(solver.h)
        units = {0: B * N, 1: B * N, 2: B * (N - 1), 3: B * (N - 1), 4: B * (N - 1), 5: B * N, 6: B * N}
        riccati_ids = (2, 4)
        desc = "iiwa14 UnParNMPCSolver N=%d T=%.2f FP64 (SURVEY 8f row 2; backward correction sweeps in place of the Riccati sweeps); " % (N, T)
        assert solver.update(0.0, q0, v0) == 0
    else:
        # BASELINE.json configs[1]: iiwa14 UnOCPSolver, N=100, T=5, FP64
        B = args.batch or 16384
        model = iiwa14_model()
        nq, nv = model.nq, model.nv
        v0 = np.zeros((B, nv))
        if args.workload == "iiwa14_task_space":
            # examples/iiwa14/task_space_ocp.cpp (TimeVaryingTaskSpace6DCost on the end-effector frame) as data
            from idocp_amd.workloads import task_circle_refs, task_space_problem
            for k in range(nv):
                model.u_max[k], model.v_max[k] = 50.0, np.pi / 2
            cost, cons = task_space_problem(model, dim=6, time_varying=True)
            q0 = np.ascontiguousarray(np.array([0, np.pi / 2, 0, np.pi / 2, 0, np.pi / 2, 0.0]) + 0.05 * rng.uniform(-1, 1, (B, nv)))
        else:
            cost, cons = unocp_problem(model)
            q0 = np.ascontiguousarray(2.0 + 0.1 * rng.uniform(-1, 1, (B, nv)))
        solver = HipUnOCP(model, cost, cons, T, N, batch=B, device=local_rank)
        if args.workload == "iiwa14_task_space":
            solver.set_task_refs(task_circle_refs(0.0, T / N, N))
        solver.set_solution_batch("q", q0)
        solver.set_solution("v", v0[0])
        KERNELS = KERNELS_UN
        launch, sync_fn, stream = lib.idocp_unocp_launch_kernel, lib.idocp_unocp_synchronize, lib.idocp_unocp_stream(solver.h)
        units = {0: B * N, 1: B * N, 2: B * N, 3: B * (N + 1), 4: B * N, 5: B * (N + 1)}
        riccati_ids = (1, 2)
        desc = "iiwa14 UnOCPSolver N=%d T=%.2f FP64 (BASELINE.json configs[1]); " % (N, T)
        if args.workload == "iiwa14_task_space":
            desc = "iiwa14 UnOCPSolver N=%d T=%.2f FP64 with a TimeVaryingTaskSpace6DCost (SURVEY 8f row 3; examples/iiwa14/task_space_ocp.cpp); " % (N, T)
    d_q, d_v = C.c_void_p(), C.c_void_p()
    capi.check(lib.idocp_device_alloc(C.byref(d_q), q0.nbytes))
    capi.check(lib.idocp_device_alloc(C.byref(d_v), v0.nbytes))
    capi.check(lib.idocp_device_upload(d_q, q0.ctypes.data, q0.nbytes))
    capi.check(lib.idocp_device_upload(d_v, v0.ctypes.data, v0.nbytes))

    def step(events=None):
        for kid in range(len(KERNELS)):
            if events is not None:
                hip.record(events[kid], stream)
            capi.check(launch(solver.h, KIDS[kid] if KIDS else kid, d_q, d_v), KERNELS[kid])
        if events is not None:
            hip.record(events[len(KERNELS)], stream)

    def sync():
        capi.check(sync_fn(solver.h))
        hip.rt.hipDeviceSynchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    ev = [[hip.event() for _ in range(len(KERNELS) + 1)] for _ in range(args.steps)]
    el = run_timed(step, sync, args.steps, args.warmup, dist, "cuda", ev)

    # per-kernel average durations from the events of the timed region, and the per-step HIP-event time of every timed step (median and
    # spread: a slow box moves all steps alike, a regression moves the median against the committed profile; boxes of this pool differ by ~5 %)
    kms = np.zeros(len(KERNELS))
    step_ev = np.zeros(args.steps)
    for k in range(args.steps):
        for kid in range(len(KERNELS)):
            kms[kid] += hip.elapsed_ms(ev[k][kid], ev[k][kid + 1])
        step_ev[k] = hip.elapsed_ms(ev[k][0], ev[k][len(KERNELS)])
    kms /= args.steps
    dom = int(np.argmax(kms))
    a_stage = A_STAGE[args.workload]
    alg_bytes = a_stage * units[dom]
    achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel: measured offline with rocprofv3 --pmc (quoted_traffic above)
    traffic, traffic_source = quoted_traffic(args.workload, B, N, KERNELS[dom])
    # the same roofline priced class by class (ANYmal chains: a_stage_of() per stage of the chain) and with SURVEY's own unit count
    # (N + 1 grid stages at the nf = 12 figure) -- `achieved` / `frac` above keep the chain-unit convention of rounds 1 - 5
    by_class = None
    if args.workload in ("anymal_trotting", "anymal_running", "anymal"):
        ch = solver.chain(0.0)
        per_chain = sum(a_stage_of(c["kind"], c["dimf"], c.get("sw_dimi", 0)) for c in ch)
        by_class = {"bytes_per_iteration_by_stage_class": per_chain, "bytes_per_iteration_survey_N_plus_1": a_stage * (N + 1),
                    "bytes_per_iteration_chain_units_at_nf12": a_stage * len(ch),
                    "dominant_kernel_frac_by_stage_class": B * per_chain / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "dominant_kernel_frac_survey_N_plus_1": B * a_stage * (N + 1) / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # parity guard inside the bench: the timed state must still be a valid solver state
    kkt = solver.kkt_error(0.0, q0, v0)
    assert np.isfinite(kkt).all(), "non-finite KKT error after the timed region"

    latency = None
    if rank == 0 and world == 1 and args.workload in ("anymal_trotting", "anymal", "anymal_running") and not args.no_latency:
        latency = latency_mode(lib, hip, build, q0, v0)
    if rank == 0:
        ms_step = 1e3 * el / args.steps
        fused = KERNELS[riccati_ids[1]] == "ocp_forward_expand"
        out = {
            "metric": "SQP iterations/sec (whole node)", "value": whole_job_value(world, B, args.steps, el), "unit": "SQP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + "batch=%d independent OCP instances per GPU, replicas across GPUs" % B,
                       "horizon": N, "batch_per_gpu": B, "parallelism": "replicas x%d" % world,
                       # the sweeps under their own keys: with the fused forward sweep (batches of instances) the second kernel is S4 + K6 + the
                       # step-size reduction in one walk, so the PAIR is not the "ms / Riccati sweep" of the metric -- that figure (S3 + S4 of one
                       # instance, comparable with cpu_baseline.ms_per_riccati_sweep) is config.latency.ms_per_riccati_sweep
                       "ms_backward_sweep": float(kms[riccati_ids[0]]),
                       ("ms_forward_sweep_with_expansion" if fused else "ms_forward_sweep"): float(kms[riccati_ids[1]]),
                       "ms_per_riccati_sweep": None if fused else float(sum(kms[i] for i in riccati_ids)),
                       "kernel_ms": {KERNELS[i]: float(kms[i]) for i in range(len(KERNELS))},
                       "step_ms_hip_events": {"median": float(np.median(step_ev)), "min": float(step_ev.min()), "max": float(step_ev.max()),
                                              "mean": float(step_ev.mean()), "note": "first to last event of each of the %d timed steps" % args.steps},
                       "max_kkt_error_after": float(np.max(kkt)), "latency": latency},
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": float(kms[dom]),
                         # the whole iteration's compulsory bytes over the whole step: one unit per stage of the CHAIN (event stages
                         # included -- 120 for the default workload, not N + 1 = 101), i.e. the largest per-launch unit count
                         "whole_step_frac": a_stage * max(units.values()) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "by_stage_class": by_class},
        }
        # the FP64 roofline of the same kernel: exact FLOPs of the reference's formulation (oracle counting build) over the launch time
        frec = oracle_flops_record(args.workload, max(units.values()) // B)
        if frec is not None:
            kf = kernel_flops(frec, KERNELS[dom])
            if kf:
                tfl = B * kf / (kms[dom] * 1e-3) / 1e12
                out["roofline"]["flops"] = {"per_launch": B * kf, "achieved_tflops": tfl, "peak": FP64_PEAK_TFLOPS, "frac": tfl / FP64_PEAK_TFLOPS,
                                            "whole_step_tflops": B * frec["flop_per_iteration"] / (ms_step * 1e-3) / 1e12,
                                            "whole_step_frac": B * frec["flop_per_iteration"] / (ms_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                            "source": "tests/golden/oracle_flops.json: exact add + mul + div + sqrt counts of the CPU restatement's formulation "
                                                      "(dense blocks, FMA = 2), regions of this kernel: " + ", ".join(frec["kernel_regions"].get(KERNELS[dom], []))}
                sq, sq_source = quoted_sq(args.workload, B, N, KERNELS[dom])
                if sq is not None:
                    # lane utilisation: useful FP64 lane-operations (FMA-equivalents = FLOP / 2) over the lane slots the kernel issued on the
                    # vector and matrix pipes (SQ_INSTS_VALU x 64; an FP64 MFMA 4x4x4_4b does 256 multiply-adds)
                    out["roofline"]["flops"]["lane_utilisation"] = (B * kf / 2) / max(1.0, sq["valu_lane_slots_per_launch"])
                    out["roofline"]["mfma_util"] = sq.get("mfma_util")
                    out["roofline"]["sq"] = sq
                out["roofline"]["sq_source"] = sq_source
        if KERNELS[dom] == "ocp_condense":
            # what the HBM fraction above does NOT say: the condensation kernel does not wait for these bytes; it is a latency / issue-bound
            # FP64 kernel -- read roofline.flops and roofline.sq next to it
            out["roofline"]["note"] = ("algorithmic bytes against the HBM peak; the kernel itself is latency / issue bound at four resident workgroups per CU "
                                       "(DESIGN.md 3.2, 7-1): roofline.flops is its FP64 roofline, roofline.sq its issue counters")
        if not args.no_cpu_baseline and world == 1:              # a reported baseline: rank 0 of the one-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.workload, model, cost, cons, T, N, q0[0], v0[0], pts, nimp=nimp)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
