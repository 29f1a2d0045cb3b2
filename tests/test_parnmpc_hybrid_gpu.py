"""GPU parity of ParNMPCSolver on horizons with discrete events (ParNMPCDiscretizer chain: aux / impulse / lift stages of the
backward-Euler formulation) against the oracle, stage by stage along the chain.  Bar: 1e-10 on the Newton direction (FP64)."""
import numpy as np
import pytest

from helpers import (ANYMAL_Q_STANDING, OCP_DIR_FIELDS, HipParNMPC, OracleParNMPC, anymal_contact_points, anymal_model, anymal_problem,
                     rel_err)

pytestmark = pytest.mark.gpu


def make_pair(N, T, events, batch=1):
    m = anymal_model()
    cost, cons = anymal_problem(m, trotting_ref=False)
    o = OracleParNMPC(m, cost, cons, T, N, max_num_impulse=3)
    g = HipParNMPC(m, cost, cons, T, N, batch=batch, max_num_impulse=3)
    q, v = ANYMAL_Q_STANDING.copy(), np.zeros(m.nv)
    for s in (o, g):
        pts = anymal_contact_points(m).copy()
        s.set_contact_status([1, 1, 1, 1], pts)
        for status, t_ev in events:
            s.push_back_contact_status(status, pts, t_ev)
        s.set_solution("q", q)
        s.set_solution("v", v)
        s.set_solution("f", [0, 0, 0.25 * (-m.total_mass * m.gravity[2])])
        s.init(0.0)
    return m, o, g, q, v


def check_chain(o, g):
    co, cg = o.chain(0.0), g.chain(0.0)
    assert len(cg) == len(co) + 1 and cg[-1]["kind"] == "terminal"           # the GPU chain ends with a placeholder
    for a, b in zip(co, cg[:-1]):
        assert (a["kind"] if a["kind"] != "terminal" else "stage") == b["kind"] and a["dimf"] == b["dimf"] and abs(a["dt"] - b["dt"]) < 1e-15
    return len(co)


LIFT = [([0, 1, 1, 0], 0.52)]
LIFT_TOUCH = [([0, 1, 1, 0], 0.52), ([1, 1, 1, 1], 0.83)]


@pytest.mark.parametrize("events", [LIFT, LIFT_TOUCH], ids=["lift", "lift+impulse"])
def test_first_iteration_direction_parity_along_the_chain(events):
    m, o, g, q, v = make_pair(20, 1.0, events)
    M = check_chain(o, g)
    e_o, e_g = o.kkt_error(0.0, q, v), g.kkt_error(0.0, q, v)
    assert abs(e_g[0] - e_o) <= 1e-9 * e_o, (e_g, e_o)
    assert o.update(0.0, q, v) == 0 and g.update(0.0, q, v) == 0
    for f in list(OCP_DIR_FIELDS) + ["dxi"]:
        e = rel_err(g.get_chain(f, M + 1)[:M], o.get_chain(f, M))
        assert e < 1e-10, (f, e)
