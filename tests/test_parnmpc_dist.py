"""Horizon-sharded ParNMPC on CPU: two gloo processes, each owning half of the stages, drive the halo protocol of
tests/parnmpc_dist.py with the oracle as the shard backend; the result must equal the single-process oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from helpers import ANYMAL_Q_STANDING, OracleParNMPCShard, anymal_contact_points, anymal_model, anymal_problem
from parnmpc_dist import ShardedParNMPC
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N, T, iters = 20, 0.5, 6
m = anymal_model()
cost, cons = anymal_problem(m, trotting_ref=False)
q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
qq = q.copy(); qq[7:] += 0.05
shard = OracleParNMPCShard(m, cost, cons, T, N, rank, world, qq, v)
shard.o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
shard.o.set_solution("q", q); shard.o.set_solution("v", v)
shard.o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
drv = ShardedParNMPC(shard, dist, rank, world)
drv.init_backward_correction(0.0)
shard.o.lib.oracle_parnmpc_init_constraints_only(shard.o.h, 0.0)
errs = []
for it in range(iters):
    drv.update(0.0)
    errs.append(float(drv.kkt_error(0.0)[0]))
out = {"rank": rank, "errs": errs, "q": shard.o.get("q").tolist(), "u": shard.o.get("u").tolist(), "lmd": shard.o.get("lmd").tolist()}
print("RESULT" + json.dumps(out), flush=True)
dist.destroy_process_group()
""" % (ROOT, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_shards_equal_the_single_process_oracle(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ANYMAL_Q_STANDING, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT")][-1][6:]))
    outs.sort(key=lambda o: o["rank"])
    # single-process reference
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    o = OracleParNMPC(m, cost, cons, 0.5, 20)
    o.set_contact_status([1, 1, 1, 1], anymal_contact_points(m))
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init(0.0)
    qq = q.copy()
    qq[7:] += 0.05
    errs = []
    for it in range(6):
        assert o.update(0.0, qq, v) == 0
        errs.append(o.kkt_error(0.0, qq, v))
    assert np.allclose(outs[0]["errs"], errs, rtol=1e-9, atol=1e-12) and np.allclose(outs[1]["errs"], errs, rtol=1e-9, atol=1e-12)
    for f in ("q", "u", "lmd"):
        both = np.concatenate([np.array(outs[0][f]), np.array(outs[1][f])])
        assert np.abs(both - o.get(f)).max() < 1e-9, f


WORKER_HYBRID = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from helpers import ANYMAL_Q_STANDING, OracleParNMPCShard, anymal_contact_points, anymal_model, anymal_problem
from parnmpc_dist import ShardedParNMPC
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N, T, iters = 20, 1.0, 8
m = anymal_model()
cost, cons = anymal_problem(m, trotting_ref=False)
q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
shard = OracleParNMPCShard(m, cost, cons, T, N, rank, world, q, v, max_num_impulse=3)
pts = anymal_contact_points(m)
shard.o.set_contact_status([1, 1, 1, 1], pts)
shard.o.push_back_contact_status([0, 1, 1, 0], pts, 0.27)       # the lift stage sits in the first shard
shard.o.push_back_contact_status([1, 1, 1, 1], pts, 0.83)       # the aux / impulse pair in the second
shard.o.set_solution("q", q); shard.o.set_solution("v", v)
shard.o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
drv = ShardedParNMPC(shard, dist, rank, world)
drv.init_backward_correction(0.0)
shard.o.lib.oracle_parnmpc_init_constraints_only(shard.o.h, 0.0)
errs = []
for it in range(iters):
    drv.update(0.0)
    errs.append(float(drv.kkt_error(0.0)[0]))
ch = shard.o.chain(0.0)
M = len(ch)
out = {"rank": rank, "errs": errs, "kinds": "".join(c["kind"][0] for c in ch), "q": shard.o.get_chain("q", M).tolist(),
       "lmd": shard.o.get_chain("lmd", M).tolist(), "f": shard.o.get_chain("f", M).tolist()}
print("RESULT" + json.dumps(out), flush=True)
dist.destroy_process_group()
""" % (ROOT, ROOT)


def test_two_shards_of_a_chain_with_events_equal_the_single_process_oracle(tmp_path):
    """A horizon with a lift and an impulse event, the chain cut between the grid stages 9 and 10: every shard discretises the
    whole horizon, keeps its slice of the chain (event stages stay with the grid stage they precede) and talks to its neighbour
    through the same halos as on an event-free horizon."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ANYMAL_Q_STANDING, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem
    script = tmp_path / "worker_hybrid.py"
    script.write_text(WORKER_HYBRID)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT")][-1][6:]))
    outs.sort(key=lambda o: o["rank"])
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    o = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
    pts = anymal_contact_points(m)
    o.set_contact_status([1, 1, 1, 1], pts)
    o.push_back_contact_status([0, 1, 1, 0], pts, 0.27)
    o.push_back_contact_status([1, 1, 1, 1], pts, 0.83)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init(0.0)
    errs = []
    for it in range(8):
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    ch = o.chain(0.0)
    kinds = "".join(c["kind"][0] for c in ch)
    assert outs[0]["kinds"] + outs[1]["kinds"] == kinds and "l" in outs[0]["kinds"] and "ai" in outs[1]["kinds"]
    assert np.allclose(outs[0]["errs"], errs, rtol=1e-8, atol=1e-12) and np.allclose(outs[1]["errs"], errs, rtol=1e-8, atol=1e-12)
    M = len(ch)
    for f in ("q", "lmd", "f"):
        both = np.concatenate([np.array(outs[0][f]), np.array(outs[1][f])])
        ref = o.get_chain(f, M)
        assert np.abs(both - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), f


WORKER_UN = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from helpers import OracleUnParNMPCShard, iiwa14_model, unocp_problem
from parnmpc_dist import ShardedParNMPC
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
N, T, iters = 20, 1.0, 8
m = iiwa14_model()
cost, cons = unocp_problem(m)
q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
shard = OracleUnParNMPCShard(m, cost, cons, T, N, rank, world, q, v)
shard.o.set_solution("q", q); shard.o.set_solution("v", v)
drv = ShardedParNMPC(shard, dist, rank, world)
drv.init_backward_correction(0.0)
errs = []
for it in range(iters):
    drv.update(0.0)
    errs.append(float(drv.kkt_error(0.0)[0]))
lo, hi = rank * (N // world), (rank + 1) * (N // world)
out = {"rank": rank, "errs": errs, "q": shard.o.get("q")[lo:hi].tolist(), "u": shard.o.get("u")[lo:hi].tolist(), "lmd": shard.o.get("lmd")[lo:hi].tolist()}
print("RESULT" + json.dumps(out), flush=True)
dist.destroy_process_group()
""" % (ROOT, ROOT)


def test_two_fixed_base_shards_equal_the_single_process_oracle(tmp_path):
    """UnParNMPCSolver: the same driver and protocol with the fixed-base oracle as the shard backend, two gloo processes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleUnParNMPC, iiwa14_model, unocp_problem
    script = tmp_path / "worker_un.py"
    script.write_text(WORKER_UN)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT")][-1][6:]))
    outs.sort(key=lambda o: o["rank"])
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnParNMPC(m, cost, cons, 1.0, 20)
    q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
    o.set_solution("q", q)
    o.set_solution("v", v)
    o.init(0.0)
    errs = []
    for it in range(8):
        assert o.update(0.0, q, v) == 0
        errs.append(o.kkt_error(0.0, q, v))
    assert np.allclose(outs[0]["errs"], errs, rtol=1e-9, atol=1e-12) and np.allclose(outs[1]["errs"], errs, rtol=1e-9, atol=1e-12)
    for f in ("q", "u", "lmd"):
        both = np.concatenate([np.array(outs[0][f]), np.array(outs[1][f])])
        assert np.abs(both - o.get(f)).max() < 1e-9, f
