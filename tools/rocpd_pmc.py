#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 --pmc run (ROCm 7 rocpd SQLite, view `counters_collection`).

usage: python tools/rocpd_pmc.py <results.db> [<results.db> ...]  > profiles/<name>.txt
One row per (kernel, counter): dispatches, sum and mean of the counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) read streams by 2x (MI355X_MICROARCH.md, HBM section) --
the raw value is printed, the correction is applied where the number is used (bench.py / DESIGN.md).
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:64]


def main():
    print(f"{'kernel':64s} {'counter':28s} {'dispatches':>10s} {'sum':>16s} {'mean/dispatch':>16s}")
    for db in sys.argv[1:]:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
        agg = {}
        for k, c, n, s, a in rows:
            key = (short(k), c)
            e = agg.setdefault(key, [0, 0.0])
            e[0] += n; e[1] += s
        print(f"# {db}")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: (-kv[1][1], kv[0])):
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            print(f"{k:64s} {c:28s} {n:10d} {s:16.1f} {s / n:16.3f}")


if __name__ == "__main__":
    main()
