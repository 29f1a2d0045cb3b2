#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (--kernel-trace --stats, optionally
--pmc) into the small text summary that is committed under profiles/.

usage: summarize_rocpd.py <results.db> [more.db ...] > profiles/rNN_<what>.txt
"""
import sqlite3
import sys


def short(name):
    """kernel name without the boilerplate, so that the template arguments that tell the variants apart stay visible"""
    for a, b in (("void ", ""), ("idocp_dev::", ""), ("LeggedDims<4, 3>", "D"), ("_kernel", ""), ("OcpBuffers", "B"), ("double const*", "cd*")):
        name = name.replace(a, b)
    return name[:62]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("# %s" % path)
        print("%-62s %7s %12s %12s %12s %12s %6s %6s %6s" %
              ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "vgpr", "agpr", "lds"))
        rows = cur.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
            "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc")
        for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
            print("%-62s %7d %12.3f %12.2f %12.2f %12.2f %6s %6s %6s" %
                  (short(name), n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, vg, ag, lds))
        try:
            rows = list(cur.execute(
                "select k.name, p.name, count(distinct e.event_id), count(*), sum(e.value) "
                "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                "join kernels k on k.id = e.event_id group by k.name, p.name order by k.name, p.name"))
        except sqlite3.Error as err:
            print("# no counter table: %s" % err)
            rows = []
        if rows:
            # one row per (kernel, counter): the counter summed over the samples of a dispatch (one per XCD / SE), averaged over dispatches
            print("\n%-62s %-24s %10s %8s %18s" % ("kernel", "counter", "dispatches", "samples", "sum_per_dispatch"))
            for kname, cname, nd, n, tot in rows:
                print("%-62s %-24s %10d %8d %18.1f" % (short(kname), cname, nd, n, tot / max(nd, 1)))
        print()


if __name__ == "__main__":
    main()
