"""The oracle's exact FLOP counter: oracle/flops.hpp + tests/golden/gen_oracle_flops.py -> tests/golden/oracle_flops.json.

* the committed file equals a live recount of the two small workloads (the counter is deterministic: integers, to the unit);
* the counting build computes the same iteration as the FP64 oracle (it is the same source with a counting scalar);
* the counts stand next to SURVEY 8(d)'s estimates (iiwa14 ~3.0e4, ANYmal nf = 12 ~5.0e5 FLOP per stage) within a factor of two."""
import json
import os
import sys

import numpy as np

from helpers import GOLDEN, OCP_DIR_FIELDS

sys.path.insert(0, GOLDEN)
import gen_oracle_flops as G      # noqa: E402


def committed():
    with open(os.path.join(GOLDEN, "oracle_flops.json")) as f:
        return json.load(f)


def test_committed_counts_equal_a_live_recount():
    ref = committed()
    lib = G.flops_lib()
    try:
        for key, (wl, N) in {"iiwa14_N20": ("iiwa14", 20), "anymal_N8": ("anymal", 8)}.items():
            live = G.count(wl, N, lib=lib)
            assert live["regions"] == ref["workloads"][key]["regions"], key
            assert live["flop_per_iteration"] == ref["workloads"][key]["flop_per_iteration"]
    finally:
        G.release_flops_lib()


def test_counting_build_computes_the_same_iteration():
    lib = G.flops_lib()
    try:
        o, q, v, _, _ = G.make_solver("anymal", 6)
        assert o.update(0.0, q, v) == 0
        counted = {f: o.get(f) for f in OCP_DIR_FIELDS}
        del o
    finally:
        G.release_flops_lib()
    o, q, v, _, _ = G.make_solver("anymal", 6)
    assert o.update(0.0, q, v) == 0
    for f in OCP_DIR_FIELDS:
        a, b = counted[f], o.get(f)
        assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b))), f


def test_counts_against_the_survey_estimates():
    ref = committed()
    cls = ref["per_stage_by_class"]
    iiwa, a12, a6 = (sum(cls[k].values()) for k in ("iiwa14", "anymal_nf12", "anymal_nf6"))
    assert 0.5 * 3.0e4 <= iiwa <= 2.0 * 3.0e4                       # ~3.0e4 FLOP per stage
    assert 0.5 * 5.0e5 <= a12 <= 2.0 * 5.0e5                        # ~5.0e5 FLOP per stage at nf = 12
    assert a6 < a12
    # the rows SURVEY itemises: a2 ~5e4, a8 ~5e4, a12 ~2.0e5, a18 ~1.6e5
    for region, est in (("rnea_derivatives", 5e4), ("mjtjinv", 5e4), ("condense", 2.0e5), ("riccati_backward", 1.6e5)):
        assert 0.5 * est <= cls["anymal_nf12"][region] <= 2.0 * est, (region, cls["anymal_nf12"][region])
    # every kernel named in kernel_regions is priced from regions that exist
    names = set()
    for w in ref["workloads"].values():
        names |= set(w["regions"])
    for regs in ref["kernel_regions"].values():
        assert set(regs) <= names | {"switching_constraint"}
