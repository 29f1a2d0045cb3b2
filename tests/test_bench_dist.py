"""Multi-process scaffolding of bench.py on CPU (gloo, world size 2).

The Riccati path does not shard: N ranks run N independent replicas and the only
collectives are the barriers around the timed region and the MAX-reduction of the elapsed time.
This test runs exactly that code (`bench.init_distributed`, `bench.run_timed`,
`bench.whole_job_value`) with a stand-in step function, one process per rank.
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist = bench.init_distributed("gloo", 0)
calls = {"step": 0, "sync": 0}
def step(events):
    calls["step"] += 1
    time.sleep(0.01 * (1 + 2 * rank))        # rank 1 is three times slower
def sync():
    calls["sync"] += 1
el = bench.run_timed(step, sync, steps=5, warmup=2, dist=dist, device="cpu")
val = bench.whole_job_value(world, 8, 5, el)
print(json.dumps({"rank": rank, "elapsed": el, "value": val, "steps": calls["step"], "syncs": calls["sync"]}), flush=True)
dist.destroy_process_group()
""" % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_replica_timing_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
    # every rank ran warmup + steps, and reports the SAME (max over ranks) elapsed time
    assert all(o["steps"] == 7 for o in outs)
    assert abs(outs[0]["elapsed"] - outs[1]["elapsed"]) < 1e-12
    assert outs[0]["elapsed"] >= 5 * 0.03 * 0.9          # bounded below by the slow rank
    # whole-job value = instances of all ranks / max time
    assert abs(outs[0]["value"] - 2 * 8 * 5 / outs[0]["elapsed"]) < 1e-9


def test_whole_job_value_weak_scaling():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.whole_job_value(1, 1024, 10, 0.5) == 1024 * 10 / 0.5
    assert bench.whole_job_value(8, 1024, 10, 0.5) == 8 * bench.whole_job_value(1, 1024, 10, 0.5)


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` (no launcher, no RANK in the environment) must start two ranks itself, rendezvous,
    and print ONE JSON line with n_gpus == 2 (gloo + a stand-in step: no GPU here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1", IDOCP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["ms_per_step"] >= 4.0 * 0.9            # bounded below by the slower rank (2 x 2 ms)
    assert abs(out["value"] - 2 * 8 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]


def test_bench_rejects_mismatched_world():
    """Started under a launcher with a world size that contradicts --gpus: fail loudly instead of printing n_gpus of the flag."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", IDOCP_BENCH_STUB="1", IDOCP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_parnmpc_gpus2_plumbing_on_gloo():
    """`python bench.py --workload anymal_parnmpc --gpus 2` end to end with the library calls stubbed (IDOCP_BENCH_STUB): the launcher
    starts two ranks, the 128-byte communicator id made on rank 0 reaches rank 1 intact over gloo (the stub checks its bytes in
    idocp_comm_init_rank), both ranks step in lock-step, ONE JSON line with strong scaling and 128 stages per rank comes out."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "anymal_parnmpc", "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["stages_per_rank"] == 128
    assert out["config"]["parallelism"] == "horizon shards x2"


def test_bench_parnmpc_gpus4_plumbing_on_gloo():
    """The same with FOUR ranks (what the driver's scaling run does next: N = 1, 2, 4, 8): every rank past 0 receives the id, the
    horizon splits into 64 stages per rank, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "anymal_parnmpc", "--gpus", "4", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["config"]["stages_per_rank"] == 64
    assert out["config"]["parallelism"] == "horizon shards x4"


def test_bench_replicas_gpus4_on_gloo():
    """`python bench.py --gpus 4` on the default (replica) workload: four ranks, one line, whole-job value = 4 x per-rank batches."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1", IDOCP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "weak"


def test_unique_id_exchange_is_a_no_op_on_one_rank():
    sys.path.insert(0, ROOT)
    import bench
    lib = bench._StubDistLib()
    raw = bench.exchange_unique_id(lib, None, 0, 1)
    assert lib.idocp_comm_init_rank(raw, 0, 1, 0, None) == 0


def test_bench_gpus2_first_contact_hang_is_ended_by_the_launcher_timeout():
    """A multi-rank run whose ranks never finish (here: rank code that sleeps inside its second step, standing in for an RCCL first
    contact that never completes) must not eat the driver's budget: `--timeout` ends the ranks the launcher started (its own process
    group), the command exits 124, and stderr carries the notice.  No JSON line is printed."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1", IDOCP_BENCH_STUB_HANG="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "anymal_parnmpc", "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--timeout", "25"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 120
    assert "did not finish within 25 s" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus2_sigterm_to_the_launcher_ends_the_ranks():
    """The launcher puts its ranks in a session of their own (so that its timeout can end exactly them); a SIGTERM aimed at the LAUNCHER
    -- a driver ending the command -- must therefore be forwarded: no rank may survive it (advisor, round 5).  The ranks hang inside their
    second step; the launcher is told to terminate; afterwards no process of its group is left and the exit code is 128 + SIGTERM."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IDOCP_BENCH_STUB="1", IDOCP_BENCH_STUB_HANG="1")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "anymal_parnmpc", "--gpus", "2", "--steps", "3", "--warmup", "1",
                             "--timeout", "600"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def descendants(pid):
        """every process below `pid` in the parent tree (launcher -> torchrun -> ranks; torchrun gives its workers sessions of their own)"""
        par = {}
        for d in os.listdir("/proc"):
            if not d.isdigit():
                continue
            try:
                st = open("/proc/%s/stat" % d).read()
                par[int(d)] = int(st[st.rindex(")") + 2:].split()[1])
            except (OSError, ValueError):
                continue
        out, frontier = [], [pid]
        while frontier:
            nxt = [p for p, pp in par.items() if pp in frontier]
            out += nxt
            frontier = nxt
        return out

    # wait until the ranks exist (launcher -> torchrun -> 2 ranks)
    deadline = time.time() + 120
    members = []
    while time.time() < deadline:
        members = descendants(proc.pid)
        if len(members) >= 3:
            break
        assert proc.poll() is None, proc.stderr.read()[-2000:]
        time.sleep(0.5)
    assert len(members) >= 3, members
    time.sleep(3.0)                                  # (let the ranks reach the hanging step)
    proc.send_signal(signal.SIGTERM)
    try:
        rc = proc.wait(timeout=60)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert rc == 128 + signal.SIGTERM, (rc, proc.stderr.read()[-2000:])
    alive = [p for p in members if os.path.exists("/proc/%d" % p) and open("/proc/%d/stat" % p).read().split(")")[-1].split()[0] != "Z"]
    assert not alive, "ranks outlived their launcher: %r" % alive


def test_parnmpc_scaling_model_of_the_bench_line():
    """config.model_ms: DESIGN section 5's step-time model, from this run's kernel times on one GPU, from the committed figures on more."""
    sys.path.insert(0, ROOT)
    import bench
    ker = {"ocp_condense": 2.0, "parnmpc_kkt_inverse": 2.0, "parnmpc_backward_serial": 0.25, "parnmpc_forward_serial": 0.75, "ocp_expand_primal": 1.0}
    m = bench.parnmpc_model_ms("anymal_parnmpc", 1, ker)
    assert abs(m["stage_parallel_ms"] - 5.0) < 1e-12 and abs(m["sweeps_ms"] - 1.0) < 1e-12
    assert abs(m["ms_per_step_by_gpus"]["1"] - 6.0) < 1e-12 and abs(m["ms_per_step_by_gpus"]["8"] - (5.0 / 8 + 1.0 + 14 * 0.015 + 0.1)) < 1e-12
    m8 = bench.parnmpc_model_ms("anymal_parnmpc", 8)
    assert m8["this_run"] == m8["ms_per_step_by_gpus"]["8"] and "committed" in m8["source"]
