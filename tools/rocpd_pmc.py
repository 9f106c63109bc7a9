#!/usr/bin/env python
"""Per-kernel PMC totals from a rocprofv3 --pmc run (ROCm 7 rocpd SQLite, view `counters_collection`).

usage: python tools/rocpd_pmc.py <results.db> [<results.db> ...]  > profiles/<name>.txt
One row per (kernel, counter): dispatches, sum and mean of the counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) read streams by 2x (MI355X_MICROARCH.md, HBM section) --
the raw value is printed, the correction is applied where the number is used (bench.py / DESIGN.md).
"""
import json
import re
import sqlite3
import sys

N_SIMD, N_SE, NOMINAL_GHZ = 1024, 32, 2.4     # MI355X: 256 CUs x 4 SIMDs, 32 shader engines (8 CUs each), MI355X_MICROARCH.md clock
GEMM_KERNELS = ("gemm_", "lstm_fwd", "lstm_bwd", "what_head")


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:64]


def mfma_util(db, digest=None, shape=None, dtype="f32"):
    """MFMA utilisation as a NUMBER from a pass that collected SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES with
    --kernel-trace.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines, so
      util_of_busy = MFMA_BUSY / (32 SIMDs per engine x SQ_BUSY)      -- share of the time a shader engine has waves in which the
                                                                         average SIMD's matrix pipe is executing
      util_of_wall = MFMA_BUSY / (1024 x kernel duration x 2.4 GHz)   -- the same against the dispatch's wall time (kernel trace of the
                                                                         same run; equals FLOP/s / peak when every MFMA is a full-rate one)
    per kernel, means per dispatch."""
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                       "where counter_name in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES') group by kernel_name, counter_name").fetchall()
    agg = {}
    for k, c, n, v in rows:
        e = agg.setdefault(short(k), {})
        e.setdefault(c, [0, 0.0]); e[c][0] += n; e[c][1] += v
    dur = {}
    try:
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
        for k, n, t in con.execute(f"select {name_col}, count(*), sum(end - start) from kernels group by {name_col}"):
            e = dur.setdefault(short(k), [0, 0.0]); e[0] += n; e[1] += t
    except sqlite3.Error:
        pass
    out = {}
    for k, e in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in e or "SQ_BUSY_CYCLES" not in e or e["SQ_VALU_MFMA_BUSY_CYCLES"][1] <= 0:
            continue
        mf = e["SQ_VALU_MFMA_BUSY_CYCLES"][1] / e["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        sq = e["SQ_BUSY_CYCLES"][1] / e["SQ_BUSY_CYCLES"][0]
        rec = {"dispatches": e["SQ_VALU_MFMA_BUSY_CYCLES"][0], "mfma_busy_cycles": round(mf, 1), "sq_busy_cycles": round(sq, 1),
               "util_of_busy": round(mf / (N_SIMD / N_SE * sq), 5) if sq > 0 else None}
        if k in dur and dur[k][0]:
            us = dur[k][1] / dur[k][0] * 1e-3
            rec["us_per_dispatch"] = round(us, 3)
            rec["util_of_wall"] = round(mf / (N_SIMD * us * NOMINAL_GHZ * 1e3), 5)
        out[k] = rec
    gemm = {k: v for k, v in out.items() if k.startswith(GEMM_KERNELS)}
    tot_m = sum(v["mfma_busy_cycles"] * v["dispatches"] for v in gemm.values())
    tot_s = sum(v["sq_busy_cycles"] * v["dispatches"] for v in gemm.values())
    tot_w = sum(v.get("us_per_dispatch", 0.0) * v["dispatches"] for v in gemm.values())
    summary = {"kernels": "every dispatch of gemm_* / lstm_* / what_head_* in the run", "util_of_busy": round(tot_m / (N_SIMD / N_SE * tot_s), 5) if tot_s else None,
               "util_of_wall": round(tot_m / (N_SIMD * tot_w * NOMINAL_GHZ * 1e3), 5) if tot_w else None}
    return {"_note": " ".join(mfma_util.__doc__.split()), "build_digest": digest, "shape": shape, "mfma_dtype": dtype,
            "dense_kernels_total": summary, "per_kernel": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"] * kv[1]["dispatches"]))}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--mfma-json":      # rocpd_pmc.py --mfma-json <db> [digest] [dtype] [H W h w T B]
        shape = [int(x) for x in sys.argv[5:11]] if len(sys.argv) >= 11 else None
        print(json.dumps(mfma_util(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, shape, sys.argv[4] if len(sys.argv) > 4 else "f32"), indent=1))
        return
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
