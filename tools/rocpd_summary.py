#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7 rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max / share.

usage: python tools/rocpd_summary.py <results.db> [--steps N]  > profiles/<name>.txt
       python tools/rocpd_summary.py <results.db> --by-position <last kernel of a step> [--every K]  > profiles/<name>_positions.txt
         per position (by start time) inside the replayed step: mean duration and mean gap to the previous kernel's end (the
         boundary cost; NEGATIVE = the kernel started while the previous one was still running: two-lane graphs, marked '*').
         --every K: the step's last kernel occurs K times per step (the two-lane step runs its optimiser slices through the
         same kernel): every K-th occurrence closes a step.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def by_position(db, last, every=1):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = sorted(cur.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])
    names = [short(r[0]) for r in rows]
    ends = [i for i, n in enumerate(names) if n.startswith(last)]
    if every > 1:
        # the occurrence that closes a step is the one followed by the longest pause (the host's replay gap): align on it
        gaps_after = [(rows[i + 1][1] - rows[i][2]) if i + 1 < len(rows) else 0 for i in ends]
        best = max(range(every), key=lambda r: sum(gaps_after[r::every]) / max(1, len(gaps_after[r::every])))
        ends = ends[best::every]
    # periods between consecutive occurrences of the step's last kernel; keep the most common length (the graph replays)
    periods = [(a + 1, b + 1) for a, b in zip(ends[:-1], ends[1:])]
    lens = {}
    for a, b in periods:
        lens[b - a] = lens.get(b - a, 0) + 1
    P = max(lens, key=lens.get)
    periods = [(a, b) for a, b in periods if b - a == P]
    ref = names[periods[-1][0]:periods[-1][1]]
    periods = [(a, b) for a, b in periods if names[a:b] == ref]
    dur = [0.0] * P; gap = [0.0] * P; mx = [0.0] * P
    for a, b in periods:
        for k in range(P):
            d = rows[a + k][2] - rows[a + k][1]
            dur[k] += d; mx[k] = max(mx[k], d)
            gap[k] += rows[a + k][1] - rows[a + k - 1][2]
    n = len(periods)
    span = sum(rows[b - 1][2] - rows[a - 1][2] for a, b in periods) / n
    print(f"# {db}: {n} identical replays of a {P}-kernel step; mean step span {span / 1e3:.1f} us, "
          f"kernel time {sum(dur) / n / 1e3:.1f} us, gaps {sum(gap) / n / 1e3:.1f} us")
    print(f"{'pos':>3s} {'kernel':60s} {'avg_us':>8s} {'max_us':>8s} {'gap_us':>8s}")
    for k in range(P):
        mark = "*" if gap[k] / n < -50.0 else " "          # started >50 ns before the previous kernel ended: overlapped
        print(f"{k:3d} {ref[k][:60]:60s} {dur[k] / n / 1e3:8.2f} {mx[k] / 1e3:8.2f} {gap[k] / n / 1e3:8.2f} {mark}")
    if "--json" in sys.argv:      # --json <path> --digest <build digest> --shape H W h w T B: what bench.py reads for the in-graph durations
        import json
        i = sys.argv.index("--shape")
        out = {"_note": "average duration of each kernel of the replayed step graph, rocprofv3 --kernel-trace of `bench.py` (begin -> end "
                        "timestamps of the dispatches inside identical graph replays; isolated launches of the same kernels are not counted)",
               "build_digest": sys.argv[sys.argv.index("--digest") + 1], "shape": [int(x) for x in sys.argv[i + 1:i + 7]],
               "mfma_dtype": sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "f32",
               "replays": n, "kernels_per_step": P, "step_kernel_time_us": round(sum(dur) / n / 1e3, 2),
               "positions": [{"pos": k, "kernel": ref[k], "avg_us": round(dur[k] / n / 1e3, 3), "max_us": round(mx[k] / 1e3, 3)} for k in range(P)]}
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)


def main():
    db = sys.argv[1]
    if "--by-position" in sys.argv:
        every = int(sys.argv[sys.argv.index("--every") + 1]) if "--every" in sys.argv else 1
        return by_position(db, sys.argv[sys.argv.index("--by-position") + 1], every)
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
