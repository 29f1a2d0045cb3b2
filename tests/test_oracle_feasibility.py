"""isCurrentSolutionFeasible of the oracle (ocp_solver.cpp:216-248, unocp_solver.cpp:228-237, parnmpc_solver.cpp:231-273):
which stage is reported follows the time-step gating of constraints_data.hpp:18-42."""
import numpy as np

from helpers import (ANYMAL_Q_STANDING, OracleOCP, OracleParNMPC, OracleUnOCP, anymal_contact_points, anymal_model, anymal_problem,
                     iiwa14_model, trotting_sequence, unocp_problem)


def test_unocp_oracle_reports_the_first_gated_stage():
    m = iiwa14_model()
    cost, cons = unocp_problem(m)
    o = OracleUnOCP(m, cost, cons, 1.0, 20)
    o.set_solution("q", np.full(m.nv, 1.0))
    assert o.infeasible_stage() == -1
    for name, bad, stage in (("q", np.full(m.nv, 2.5), 2), ("v", np.full(m.nv, -50.0), 1), ("u", np.full(m.nv, 1.0e4), 0)):
        o = OracleUnOCP(m, cost, cons, 1.0, 20)
        o.set_solution(name, bad)
        assert o.infeasible_stage() == stage


def test_ocp_oracle_checks_cones_and_limits_along_the_chain():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=True)

    def make():
        o = OracleOCP(m, cost, cons, 1.55, 30, max_num_impulse=3)
        trotting_sequence(o, m, 2)
        o.set_solution("q", ANYMAL_Q_STANDING)
        o.set_solution("v", np.zeros(m.nv))
        o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        o.init_constraints(0.0)
        return o

    o = make()
    assert o.infeasible_stage() == -1
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for _ in range(3):
        assert o.update(0.0, q, v) == 0
        assert o.infeasible_stage() == -1
    qbad = q.copy()
    qbad[7:] = 100.0
    for name, bad, where in (("f", [1.0, 0.0, 0.1], 0), ("f", [0.0, 0.0, -1.0], 0), ("u", np.full(m.nv - 6, 1.0e4), 0),
                             ("v", np.concatenate([np.zeros(6), np.full(m.nv - 6, 1.0e3)]), 1), ("q", qbad, 2)):
        o = make()
        o.set_solution(name, bad)
        assert o.infeasible_stage() == where, name


def test_parnmpc_oracle_feasibility():
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    o = OracleParNMPC(m, cost, cons, 1.0, 20, max_num_impulse=3)
    pts = anymal_contact_points(m).copy()
    o.set_contact_status([1, 1, 1, 1], pts)
    o.push_back_contact_status([0, 1, 1, 0], pts, 0.52)
    o.set_solution("q", ANYMAL_Q_STANDING)
    o.set_solution("v", np.zeros(m.nv))
    o.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
    o.init(0.0)
    assert o.infeasible_stage() == -1
    qbad = ANYMAL_Q_STANDING.copy()
    qbad[7:] = -100.0
    o.set_solution("q", qbad)
    assert o.infeasible_stage() == 1          # stage i is created with time step i + 1: position limits from stage 1
    o.set_solution("u", np.full(m.nv - 6, 1.0e4))
    assert o.infeasible_stage() == 0
