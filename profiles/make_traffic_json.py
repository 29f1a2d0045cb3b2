#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic_<workload>.json from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).

usage: make_traffic_json.py <fetch_results.db> <write_results.db> <workload> <batch> <horizon> <round>

Counter units and the gfx950 correction (see the guide's HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE are summed over the per-XCD samples and are in kB (x1024 -> bytes);
FETCH_SIZE is doubled because gfx950 tallies its 128-B read requests at 64 B.
"""
import json
import re
import sqlite3
import sys


def per_launch(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select k.name, count(distinct e.event_id), sum(e.value) from rocpd_pmc_event e "
        "join rocpd_info_pmc p on e.pmc_id = p.id join kernels k on k.id = e.event_id "
        "where p.name = ? group by k.name", (counter,)).fetchall()
    out = {}
    for name, launches, total in rows:
        m = re.search(r"idocp_dev::(\w+?)_kernel(<[^>]*?(false|true)?[^>]*>)?", name)
        if not m:
            continue
        key = m.group(1)
        if "condense" in key and re.search(r"_kernel<[^,>]+,\s*true", name):      # RESIDUAL is the second template argument
            key += "_residual"
        elif key.startswith("un_linearize") and re.search(r"_kernel<\d+,\s*1\b", name):
            key += "_residual"
        # the template variants of one kernel that run side by side in every step (K5b per stage class, K5a regular + impulse
        # launch) share a key: bytes of all of them per step = sum of their totals / launches of the most frequent one
        cur = out.get(key, (0, 0.0))
        out[key] = (max(cur[0], launches), cur[1] + 1024.0 * total)
    return {k: v[1] / v[0] for k, v in out.items()}


def main():
    fetch_db, write_db, workload, batch, horizon, rnd = sys.argv[1:7]
    f = per_launch(fetch_db, "FETCH_SIZE")
    w = per_launch(write_db, "WRITE_SIZE")
    rec = {
        "round": int(rnd), "workload": workload, "batch": int(batch), "horizon": int(horizon),
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload %s --steps 3 --warmup 1 "
                   "--no-cpu-baseline (two separate passes)" % workload,
        "correction": "counters in kB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "
                      "HBM section); WRITE_SIZE as reported",
        "raw_bytes_per_launch": {k: {"FETCH_SIZE": f.get(k), "WRITE_SIZE": w.get(k)} for k in sorted(set(f) | set(w))},
        "hbm_bytes_per_launch": {k: 2.0 * f.get(k, 0.0) + w.get(k, 0.0) for k in sorted(set(f) | set(w))},
    }
    json.dump(rec, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
