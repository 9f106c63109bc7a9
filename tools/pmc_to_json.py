#!/usr/bin/env python
"""HBM-side bytes per launch of the ST kernels from rocprofv3 PMC passes -> the JSON bench.py reads for `roofline.traffic`.

usage: python tools/pmc_to_json.py --fetch <FETCH_SIZE.db> --write <WRITE_SIZE.db> --digest <air_build_digest> \
                                   --shape H W h w T B  > profiles/<tag>_instep_pmc.json

FETCH_SIZE / WRITE_SIZE are collected in separate passes (TCC slots) with --kernel-trace only, unit KiB.  Per
MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts a wide (16 B/lane) coalesced read stream at half its bytes:
fetch bytes = FETCH_SIZE * 1024 * 2 (an upper bound for kernels that also read narrow rows); WRITE_SIZE matched known output
bytes on this kernel family in round 1 (x1).  Values are means per dispatch over the profiled run.
"""
import argparse
import json
import re
import sqlite3

KERNELS = {                      # bench.py key -> kernel-name prefix in the trace
    "st_read_fwd": "st_read_fwd_", "st_read_bwd": "st_read_bwd_kernel",
    "canvas_unroll_fwd": "st_write_fwd_kernel", "canvas_unroll_bwd": "st_write_bwd_",     # (unit-major and image-major forms)
    "attend_fwd": "attend_fwd_kernel", "attend_bwd": "attend_bwd_kernel",
    "canvas_fused": "canvas_fused_",     # forward + recompute-form backward as one launch (latency-regime train step)
}


def means(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, s in rows:
        short = re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))
        e = out.setdefault(short, [0, 0.0]); e[0] += n; e[1] += s
    return {k: (v[1] / v[0], v[0]) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True); ap.add_argument("--write", required=True)
    ap.add_argument("--digest", required=True); ap.add_argument("--shape", type=int, nargs=6, required=True)
    a = ap.parse_args()
    f, w = means(a.fetch, "FETCH_SIZE"), means(a.write, "WRITE_SIZE")
    kernels = {}
    for key, prefix in KERNELS.items():
        fk = [k for k in f if k.startswith(prefix)]
        wk = [k for k in w if k.startswith(prefix)]
        if not fk or not wk:
            continue
        fetch_kib = sum(f[k][0] * f[k][1] for k in fk) / sum(f[k][1] for k in fk)
        write_kib = sum(w[k][0] * w[k][1] for k in wk) / sum(w[k][1] for k in wk)
        kernels[key] = {"kernel": fk[0], "dispatches": int(sum(f[k][1] for k in fk)), "FETCH_SIZE_KiB": round(fetch_kib, 3),
                        "WRITE_SIZE_KiB": round(write_kib, 3), "fetch_bytes_corrected": int(fetch_kib * 1024 * 2),
                        "write_bytes": int(write_kib * 1024),
                        "traffic_bytes": int(fetch_kib * 1024 * 2 + write_kib * 1024)}
    print(json.dumps({"_note": __doc__.strip().split("\n\n")[2].replace("\n", " "), "build_digest": a.digest,
                      "shape": a.shape, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
