#!/usr/bin/env python
"""Outcome distribution of a set of training logs (scripts/multi_mnist.py `log.jsonl`): final count accuracy / mean step count on the
held-out set per seed, best accuracy reached and when, and whether the run ended in NaN.

usage: python tools/summarize_runs.py profiles/r03_train_glyphs_300k_seed*_log.jsonl
"""
import json
import math
import re
import sys


def main():
    rows = []
    for path in sys.argv[1:]:
        recs = [json.loads(l) for l in open(path) if l.strip()]
        test = [r for r in recs if r.get("data") == "test"]
        if not test:
            continue
        last = test[-1]
        finite = [r for r in test if isinstance(r.get("loss"), float) and math.isfinite(r["loss"])]
        best = max(finite, key=lambda r: r.get("num_step_acc", 0.0)) if finite else None
        m = re.search(r"seed(\d+)", path)
        rows.append(dict(seed=int(m.group(1)) if m else -1, final_acc=last.get("num_step_acc"), final_steps=last.get("num_step"),
                         nan=not (isinstance(last.get("loss"), float) and math.isfinite(last["loss"])),
                         best_acc=best.get("num_step_acc") if best else None, best_at=best.get("step") if best else None))
    rows.sort(key=lambda r: r["seed"])
    print(f"{'seed':>4s} {'final_acc':>9s} {'final_steps':>11s} {'nan':>4s} {'best_acc':>8s} {'best_at':>8s}")
    for r in rows:
        print(f"{r['seed']:4d} {r['final_acc']:9.3f} {r['final_steps']:11.3f} {str(r['nan']):>4s} "
              f"{(r['best_acc'] if r['best_acc'] is not None else float('nan')):8.3f} {str(r['best_at']):>8s}")
    n = len(rows)
    alive = [r for r in rows if not r["nan"]]
    bins = [("acc >= 0.95", lambda a: a >= 0.95), ("0.80 <= acc < 0.95", lambda a: 0.8 <= a < 0.95), ("0.50 <= acc < 0.80", lambda a: 0.5 <= a < 0.8),
            ("0.20 <= acc < 0.50", lambda a: 0.2 <= a < 0.5), ("acc < 0.20", lambda a: a < 0.2)]
    print(f"\n{n} runs, {n - len(alive)} ended in NaN; of the {len(alive)} finite ones:")
    for name, f in bins:
        print(f"  {name:20s} {sum(1 for r in alive if f(r['final_acc']))}")
    print(f"best accuracy reached at any evaluation >= 0.95 in {sum(1 for r in rows if (r['best_acc'] or 0) >= 0.95)} runs")


if __name__ == "__main__":
    main()
