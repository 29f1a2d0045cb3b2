"""isCurrentSolutionFeasible of the three solvers through the C ABI against the oracle: same verdict and the same first
offending stage on feasible iterates, on iterates that break one constraint family at a time (the time-step gating of
constraints_data.hpp:18-42 decides WHICH stage is reported), and per instance of a batch."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, HipOCP, HipParNMPC, HipUnOCP, OracleOCP, OracleParNMPC, OracleUnOCP, anymal_contact_points,
                     anymal_model, anymal_problem, iiwa14_model, trotting_sequence, unocp_problem)

pytestmark = pytest.mark.gpu


def test_unocp_feasibility_matches_the_oracle():
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o, g = OracleUnOCP(m, cost, cons, 1.0, 20), HipUnOCP(m, cost, cons, 1.0, 20, batch=2)
    q, v = np.full(m.nv, 1.0), np.zeros(m.nv)
    for s in (o, g):
        s.set_solution("q", q)
        s.set_solution("v", v)
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]
    for _ in range(3):                                   # the interior-point iterates stay strictly inside
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]
    # one family at a time: position limits exist from stage 2, velocity limits from stage 1, torque limits from stage 0
    for name, bad, stage in (("q", np.full(m.nv, 2.5), 2), ("v", np.full(m.nv, -50.0), 1), ("u", np.full(m.nv, 1.0e4), 0)):
        o2, g2 = OracleUnOCP(m, cost, cons, 1.0, 20), HipUnOCP(m, cost, cons, 1.0, 20, batch=2)
        o2.set_solution(name, bad)
        vals = np.zeros((2, m.nv))
        vals[1] = bad
        g2.set_solution_batch(name, vals)                # instance 0 stays feasible
        assert o2.infeasible_stage() == stage
        assert list(g2.infeasible_stage()) == [-1, stage]


def hybrid_pair(batch=1):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)
    o = OracleOCP(m, cost, cons, 1.55, 30, max_num_impulse=3)
    g = HipOCP(m, cost, cons, 1.55, 30, batch=batch, max_num_impulse=3)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        trotting_sequence(s, m, 2)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init_constraints(0.0)
    return m, o, g, q, v


def test_ocp_feasibility_matches_the_oracle_along_the_chain():
    m, o, g, q, v = hybrid_pair(batch=2)
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]
    for _ in range(3):
        assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
        assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1, -1]
    qbad = q.copy()
    qbad[7:] = 100.0
    cases = (("f", [1.0, 0.0, 0.1]),        # outside the friction pyramid
             ("f", [0.0, 0.0, -1.0]),       # pulling
             ("u", np.full(m.nv - 6, 1.0e4)), ("v", np.concatenate([np.zeros(6), np.full(m.nv - 6, 1.0e3)])), ("q", qbad))
    seen = set()
    for name, bad in cases:
        m, o, g, q, v = hybrid_pair()
        for s in (o, g):
            s.set_solution(name, bad)
        w = o.infeasible_stage()
        assert w >= 0 and list(g.infeasible_stage()) == [w], (name, w, g.infeasible_stage())
        seen.add(w)
    assert seen == {0, 1, 2}                # torque / cone from stage 0, velocity from 1, position from 2


def test_parnmpc_feasibility_matches_the_oracle():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)

    def pair():
        o = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
        g = HipParNMPC(m, cost, cons, 1.0, 20, batch=1, max_num_impulse=3)
        for s in (o, g):
            pts = anymal_contact_points(m).copy()
            s.set_contact_status([1, 1, 1, 1], pts)
            s.push_back_contact_status([0, 1, 1, 0], pts, 0.52)
            s.push_back_contact_status([1, 1, 1, 1], pts, 0.83)
            s.set_solution("q", ANYMAL_Q_STANDING)
            s.set_solution("v", np.zeros(m.nv))
            s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
            s.init(0.0)
        return o, g

    o, g = pair()
    assert o.infeasible_stage() == -1 and list(g.infeasible_stage()) == [-1]
    qbad = ANYMAL_Q_STANDING.copy()
    qbad[7:] = -100.0
    for name, bad in (("f", [0.0, 2.0, 0.1]), ("u", np.full(m.nv - 6, -1.0e4)), ("q", qbad)):
        o, g = pair()
        for s in (o, g):
            s.set_solution(name, bad)
        w = o.infeasible_stage()
        assert w >= 0 and list(g.infeasible_stage()) == [w], (name, w, g.infeasible_stage())
