"""GPU parity of UnParNMPCSolver (iiwa14) against the oracle, through the C ABI: phase by phase on the first iteration,
then iterate by iterate.  Bar: 1e-10 on the Newton direction (FP64)."""
import numpy as np
import pytest

from helpers import HipUnParNMPC, OracleUnParNMPC, iiwa14_model, rel_err, unocp_problem

pytestmark = pytest.mark.gpu
TOL = 1e-10
SOL = ("q", "v", "a", "u", "lmd", "gmm", "beta")
DIR = tuple("d" + f for f in SOL)
NEW = tuple("new_" + f for f in ("lmd", "gmm", "a", "q", "v"))


def make_pair(N, T, batch=1, q0=2.0):
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o, g = OracleUnParNMPC(m, cost, cons, T, N), HipUnParNMPC(m, cost, cons, T, N, batch=batch)
    q, v = np.full(m.nv, q0), np.zeros(m.nv)
    for s in (o, g):
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.init(0.0)
    return m, o, g, q, v


def compare(o, g, fields, tol, what):
    for f in fields:
        e = rel_err(g.get(f), o.get(f))
        assert e < tol, (what, f, e)


@pytest.mark.parametrize("N,T", [(20, 1.0), (3, 0.3), (64, 2.0), (1, 0.05), (2, 0.1)])
def test_phase_by_phase_parity_of_the_first_iteration(N, T):
    m, o, g, q, v = make_pair(N, T)
    # oracle stage 0 = linearize + KKT inverse + coarse update = GPU phases 0, 1
    assert o.stage(0, 0.0, q, v) == 0
    g.phase(0, q, v)
    g.phase(1, q, v)
    compare(o, g, NEW, TOL, "coarse update")
    for k, name in ((1, "backward serial"), (2, "backward parallel"), (3, "forward serial")):
        assert o.stage(k, 0.0, q, v) == 0
        g.phase(k + 1, q, v)
        compare(o, g, NEW, TOL, name)
    assert o.stage(4, 0.0, q, v) == 0
    g.phase(5, q, v)
    compare(o, g, NEW, TOL, "forward parallel")
    compare(o, g, DIR, TOL, "direction")
    ao, bo = o.step_sizes()
    ag, bg = g.step_sizes()
    assert abs(ag[0] - ao) < 1e-9 and abs(bg[0] - bo) < 1e-9
    assert o.stage(5, 0.0, q, v) == 0
    g.phase(6, q, v)
    compare(o, g, SOL, TOL, "integrated solution")
    so, do = o.constraint_data()
    sg, dg = g.constraint_data()
    assert rel_err(sg, so) < TOL and rel_err(dg, do) < TOL


def test_multi_iteration_parity_and_convergence():
    # ocpbenchmarker::Convergence protocol of examples/iiwa14/unparnmpc_benchmark.cpp (N = 20, T = 1, 100 iterations)
    m, o, g, q, v = make_pair(20, 1.0, batch=2)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) < 1e-9 * max(1.0, e_o) and e_g[0] == e_g[1]
    for it in range(100):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        if it < 3:                                   # iterate-by-iterate parity while rounding has not yet been amplified
            compare(o, g, DIR, 1e-9 if it else TOL, "iteration %d" % it)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert e_o < 1e-6 and e_g[0] < 1e-6 and e_g[1] < 1e-6
    compare(o, g, ("q", "v", "a", "u"), 1e-6, "converged solution")
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]


def test_batch_instances_are_independent():
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    rng = np.random.default_rng(3)
    B, N = 5, 12
    q0 = 1.0 + 0.5 * rng.uniform(-1, 1, (B, m.nv))
    v0 = 0.2 * rng.uniform(-1, 1, (B, m.nv))
    g = HipUnParNMPC(m, cost, cons, 0.6, N, batch=B)
    g.set_solution_batch("q", q0)
    g.set_solution_batch("v", v0)
    g.init(0.0)
    assert g.update(0.0, q0, v0) == 0
    for b in (0, 3, 4):
        o = OracleUnParNMPC(m, cost, cons, 0.6, N)
        o.set_solution("q", q0[b])
        o.set_solution("v", v0[b])
        o.init(0.0)
        assert o.update(0.0, q0[b], v0[b]) == 0
        for f in DIR:
            assert rel_err(g.get(f, b), o.get(f)) < TOL, (b, f)


def test_handle_kinds_are_not_interchangeable():
    from helpers import HipUnOCP, P, arr
    E_ARG = -1
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    g, u = HipUnParNMPC(m, cost, cons, 1.0, 4), HipUnOCP(m, cost, cons, 1.0, 4)
    q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
    assert g.lib.idocp_unocp_update_solution(g.h, 0.0, P(arr(q)), P(arr(v)), 0) == E_ARG
    assert b"UnParNMPCSolver" in g.lib.idocp_last_error()
    assert u.lib.idocp_unparnmpc_update_solution(u.h, 0.0, P(arr(q)), P(arr(v)), 0) == E_ARG


def test_filter_line_search_parity():
    # UnLineSearch on the backward-Euler stages (src/line_search/unline_search.cpp:85-121)
    m, o, g, q, v = make_pair(20, 1.0, batch=2, q0=1.0)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    g.kkt_error(0.0, q, v)                                                   # leaves the measured state on the device
    # the state the line search sees: direction and step sizes computed, iterate not yet updated
    for k in range(5):
        assert o.stage(k, 0.0, q, v) == 0
    for ph in range(6):
        g.phase(ph, q, v)
    amax = o.step_sizes()[0]
    for alpha in (0.0, 0.5 * amax, amax):
        co, vo = o.cost_and_violation(alpha, q, v)
        cg, vg = g.cost_and_violation(alpha)
        assert abs(cg[0] - co) < 1e-10 * max(1.0, abs(co)) and abs(vg[0] - vo) < 1e-10 * max(1.0, vo), (alpha, cg, co, vg, vo)
    assert o.stage(5, 0.0, q, v) == 0
    g.phase(6, q, v)
    for it in range(8):
        assert o.update(0.0, q, v, line_search=True) == 0 and g.update(0.0, q, v, line_search=True) == 0
        ao, _ = o.step_sizes()
        ag, _ = g.step_sizes()
        assert abs(ag[0] - ao) < 1e-9 and ag[0] == ag[1], (it, ag, ao)
        compare(o, g, ("q", "v", "a", "u"), 1e-8, "line search iteration %d" % it)


@pytest.mark.parametrize("batch", [1, 4, 7])
def test_ragged_batch_sizes(batch):
    # the per-stage kernels pack 3 (K1, K9u), 8 (K11u, K3) or 2 (K10u) stages and the sweeps 4 instances per wavefront:
    # batch * N not a multiple of any of them
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    N = 5
    rng = np.random.default_rng(batch)
    q0 = 1.0 + 0.3 * rng.uniform(-1, 1, (batch, m.nv))
    v0 = 0.1 * rng.uniform(-1, 1, (batch, m.nv))
    g = HipUnParNMPC(m, cost, cons, 0.25, N, batch=batch)
    g.set_solution_batch("q", q0)
    g.set_solution_batch("v", v0)
    g.init(0.0)
    for _ in range(2):
        assert g.update(0.0, q0, v0) == 0
    b = batch - 1
    o = OracleUnParNMPC(m, cost, cons, 0.25, N)
    o.set_solution("q", q0[b])
    o.set_solution("v", v0[b])
    o.init(0.0)
    for _ in range(2):
        assert o.update(0.0, q0[b], v0[b]) == 0
    for f in DIR:
        assert rel_err(g.get(f, b), o.get(f)) < 1e-9, f


def test_two_shards_on_one_gpu_equal_the_whole_horizon():
    """Horizon sharding of the fixed-base solver without a second GPU: two shard handles of 10 stages each on this GPU, the
    halo protocol of tests/parnmpc_dist.py executed by hand in its pipeline order, against one handle of 20."""
    import torch
    from helpers import P, arr
    from idocp_amd import capi
    from parnmpc_dist import HipUnParNMPCShard
    m, o, g, q, v = make_pair(20, 1.0, q0=1.0)
    cost, cons = unocp_problem(m)
    lib = capi.lib()
    shards = [HipUnParNMPCShard(m, cost, cons, 1.0, 20, r, 2, 1, 0) for r in range(2)]
    for sh in shards:
        capi.check(lib.idocp_unocp_set_solution(sh.h, b"q", P(arr(q))))
        capi.check(lib.idocp_unocp_set_solution(sh.h, b"v", P(arr(v))))
        sh.phase("init_aux", 0.0)
    s0, s1 = shards
    s0.set_initial_state(q[None, :], v[None, :])

    def boundary():
        s1.import_(0, s0.export(0))
        s0.import_(1, s1.export(1))
        s0.import_(2, s1.export(2))

    def get(sh, name):
        out = np.zeros((11, m.nv))
        capi.check(lib.idocp_unocp_get_solution(sh.h, name.encode(), 0, P(out)))
        return out[:10]

    for it in range(5):
        assert g.update(0.0, q, v) == 0
        boundary()
        for sh in shards:
            sh.phase("linearize", 0.0)
        s1.phase("bwd_serial", 0.0)
        s0.import_(3, s1.export(3))
        s0.phase("bwd_serial", 0.0)
        for sh in shards:
            sh.phase("bwd_parallel", 0.0)
        s0.phase("fwd_serial", 0.0)
        s1.import_(4, s0.export(4))
        s1.phase("fwd_serial", 0.0)
        for sh in shards:
            sh.phase("fwd_parallel", 0.0)
        steps = torch.minimum(s0.local_steps(), s1.local_steps())
        ag, bg = g.step_sizes()
        assert abs(float(steps[0, 0]) - ag[0]) < 1e-12 and abs(float(steps[0, 1]) - bg[0]) < 1e-12
        for sh in shards:
            sh.set_steps(steps)
            sh.phase("integrate", 0.0)
        for name in SOL:
            both = np.concatenate([get(s0, name), get(s1, name)])
            assert rel_err(both, g.get(name)) < 1e-12, (it, name)
    boundary()
    e2 = s0.err2(0.0) + s1.err2(0.0)
    assert abs(float(e2[0].sqrt()) - g.kkt_error(0.0, q, v)[0]) < 1e-9 * max(1.0, float(e2[0].sqrt()))
