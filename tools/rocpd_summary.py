#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7 rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max / share.

usage: python tools/rocpd_summary.py <results.db> [--steps N]  > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    db = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# {db}: {len(rows)} kernel dispatches, {total / 1e6:.3f} ms total GPU kernel time"
          + (f", {total / steps / 1e3:.1f} us/step over {steps} steps" if steps else ""))
    print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share%':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:70s} {a[0]:8d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100.0 * a[1] / total:7.2f}")


if __name__ == "__main__":
    main()
