#!/usr/bin/env python
"""Where a long training run turns non-finite, replayed through the CPU oracle ("next" #1).

Runs on the GPU box.  Three stages, one report (JSON + a text summary):

 1. FIND   -- train the script's configuration (scripts/multi_mnist.py:24-94: batch 64, T = 3, lr 1e-4, procedural digits, the
              HBM-resident feeder, hipGraph replays) from `--seed`, snapshotting the engine state every `--chunk` updates, until a
              parameter is non-finite (or `--max-iters`).
 2. REPLAY -- restart from the snapshot >= `--back` updates before that, one replay at a time, reading back after every update
              the batch indices and the noise the graph drew (eng.eps_where / eps_what / u_pres) and a few extreme values; the
              first update after which the engine's parameters are non-finite is U*.
 3. ORACLE -- (a) free run: O.train_step in fp32 AND fp64 from the same snapshot with the SAME batches and noise; the first
              non-finite update of each is recorded next to U*.  (b) teacher forced, the last `--forced` updates up to U*: from the
              ENGINE's state before update k the fp32 / fp64 oracle takes one step; compared with the engine's state after it
              (relative error of the parameter delta while finite; at U*: which gradient tensors are non-finite on either side).

The oracle is the checker here (tools/ is not the product).  Nothing reads /root/reference.
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def unflatten(sd, key, dtype):
    out = {}
    flat = sd[key]
    for k, o in sd["param_offsets"].items():
        shp = tuple(sd["param_shapes"][k])
        n = int(np.prod(shp)) if len(shp) else 1
        out[k] = flat[o:o + n].reshape(shp).to(dtype).clone()
    return out


def oracle_state(sd, dtype):
    p = unflatten(sd, "flat_params", dtype)
    ms, mg, mom = (unflatten(sd, k, dtype) for k in ("flat_ms", "flat_mg", "flat_mom"))
    return p, {k: dict(ms=ms[k], mg=mg[k], mom=mom[k]) for k in p}


def nonfinite_names(d):
    return sorted(k for k, v in d.items() if torch.is_tensor(v) and not bool(torch.isfinite(v).all()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=30)
    ap.add_argument("--max-iters", type=int, default=300000)
    ap.add_argument("--chunk", type=int, default=250)
    ap.add_argument("--back", type=int, default=500)
    ap.add_argument("--forced", type=int, default=40)
    ap.add_argument("--after", type=int, default=5, help="updates replayed past U* (noise recorded for the oracle's free run)")
    ap.add_argument("--samples", type=int, default=60000)
    ap.add_argument("--out", default="gpurun_out/blowup")
    ap.add_argument("--oracle-threads", type=int, default=8)
    ap.add_argument("--legacy-data", action="store_true",
                    help="the round-3 dataset generator (tools/legacy/data_r03.py: PCG64 draws, two position draws per object) instead of "
                         "the reference-order one -- reproduces runs recorded before the generator was made draw-for-draw faithful")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    if args.legacy_data:
        import importlib.util
        spec = importlib.util.spec_from_file_location("data_r03", os.path.join(ROOT, "tools", "legacy", "data_r03.py"))
        legacy = importlib.util.module_from_spec(spec); spec.loader.exec_module(legacy)
        procedural_multi_mnist = legacy.procedural_multi_mnist
    else:
        from attend_infer_repeat_amd.data import procedural_multi_mnist
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    from oracle import air_oracle as O

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = 64
    ecfg, ocfg = EngineConfig(), O.AIRConfig()
    data = procedural_multi_mnist(args.samples, seed=args.seed)
    imgs = torch.from_numpy(data["imgs"].astype(np.float32) / 255.0).reshape(args.samples, -1).contiguous()
    eng = AIREngine(ecfg, B, device=dev, seed=args.seed)
    eng.init_parameters(seed=args.seed)
    eng.attach_dataset(imgs.to(dev), shuffle=True, seed=args.seed)
    eng.capture()

    # ---- 1. find ---------------------------------------------------------------------------------------------------------
    keep = args.back // args.chunk + 2
    snaps = collections.deque(maxlen=keep)
    snaps.append(eng.state_dict())
    t0 = time.time()
    bad_at = None
    while eng.global_step < args.max_iters:
        for _ in range(args.chunk):
            eng.train_step()
        eng.synchronize()
        if not bool(torch.isfinite(eng.flat_params).all()):
            bad_at = eng.global_step
            break
        snaps.append(eng.state_dict())
    report = dict(seed=args.seed, batch=B, find=dict(first_nonfinite_check=bad_at, chunk=args.chunk,
                                                     seconds=round(time.time() - t0, 1), updates=eng.global_step))
    print("find:", report["find"], flush=True)
    if bad_at is None:
        report["result"] = "no non-finite parameter within %d updates" % args.max_iters
        json.dump(report, open(os.path.join(args.out, "report_seed%d.json" % args.seed), "w"), indent=1)
        return 0
    s0 = snaps[0]
    start = int(s0["global_step"])
    torch.save(s0, os.path.join(args.out, "snapshot_seed%d_step%d.pt" % (args.seed, start)))

    # ---- 2. replay -------------------------------------------------------------------------------------------------------
    def replay(n_max, stop_after_bad, want_states_from=None):
        eng.load_state_dict(s0)
        rec, states = [], {}
        u_star, k = None, start
        while k < start + n_max:
            if want_states_from is not None and k >= want_states_from:
                states[k] = eng.state_dict()
            eng.train_step(); eng.synchronize()
            k += 1
            o = eng.outputs()
            fin = bool(torch.isfinite(eng.flat_params).all())
            rec.append(dict(
                update=k, idx=eng.batch_idx.cpu().clone(), eps_where=eng.eps_where.cpu().clone(), eps_what=eng.eps_what.cpu().clone(),
                u_pres=eng.u_pres.cpu().clone(),
                stats=dict(where_scale_min=float(o["where_scale"].min()), where_scale_max=float(o["where_scale"].max()),
                           where_abs_max=float(o["where"].abs().max()), kl_where=float(o["kl_where"]), kl_what=float(o["kl_what"]),
                           rec=float(o["rec_loss"]), opt_loss=float(o["opt_loss"]), num_step=float(o["num_step_per_sample"].mean()),
                           grads_finite=bool(torch.isfinite(eng.flat_grads).all()), params_finite=fin,
                           grad_abs_max=float(eng.flat_grads.abs().max()))))
            if not fin and u_star is None:
                u_star = k
                if want_states_from is not None:
                    states[k] = eng.state_dict()
                    states["grads_at_%d" % k] = {n: g.detach().cpu().clone() for n, g in eng.named_grads().items()}
                    # where the non-finite values FIRST appear in the step's intermediates (every buffer the engine holds)
                    where_rows = eng.where.reshape(-1, 4)
                    diag = {}
                    for name, t in sorted(eng._bufs.items()):
                        if t.is_floating_point():
                            bad = ~torch.isfinite(t)
                            if bool(bad.any()):
                                diag[name] = dict(nonfinite=int(bad.sum()), shape=list(t.shape))
                                if t.dim() == 2 and t.shape[0] == where_rows.shape[0]:
                                    rows = bad.any(1).nonzero().reshape(-1)[:6]
                                    diag[name]["rows"] = rows.tolist()
                                    diag[name]["where_of_rows"] = where_rows[rows].tolist()
                                    diag[name]["values"] = t[rows][:, :12].tolist()
                                    diag[name]["presence_of_rows"] = eng.presence.reshape(-1)[rows].tolist()
                    states["nonfinite_buffers_at_%d" % k] = diag
            if u_star is not None and k >= u_star + stop_after_bad:
                break
        return rec, u_star, states

    rec, u_star, _ = replay(bad_at - start + args.after, args.after)
    assert u_star is not None, "replay from the snapshot stayed finite: the run is not bitwise reproducible"
    report["replay"] = dict(snapshot_update=start, first_nonfinite_update=u_star,
                            engine_stats_last=[dict(update=r["update"], **r["stats"]) for r in rec[-(args.after + 12):]])
    print("replay: snapshot %d, engine parameters non-finite after update %d" % (start, u_star), flush=True)
    first_bad_grad = next((r["update"] for r in rec if not r["stats"]["grads_finite"]), None)
    first_inf_loss = next((r["update"] for r in rec if not np.isfinite(r["stats"]["opt_loss"])), None)
    report["replay"].update(first_nonfinite_gradient=first_bad_grad, first_nonfinite_loss=first_inf_loss)

    def batch_of(r, dtype):
        obs = imgs[r["idx"]].reshape(B, *ocfg.img_size).to(dtype)
        T = ocfg.max_steps
        noise = dict(eps_where=r["eps_where"].reshape(T, B, 4).to(dtype), eps_what=r["eps_what"].reshape(T, B, -1).to(dtype),
                     u_pres=r["u_pres"].reshape(T, B, 1).to(dtype))
        return obs, noise

    # ---- 3a. free-running oracle -----------------------------------------------------------------------------------------
    torch.set_num_threads(args.oracle_threads)
    free = {}
    for dtype, name in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        p, slots = oracle_state(s0, dtype)
        t1 = time.time()
        first = dict(loss=None, grads=None, params=None)
        trace = []
        for r in rec:
            obs, noise = batch_of(r, dtype)
            res, grads = O.train_step(p, slots, ocfg, obs, noise, global_step=r["update"] - 1)
            st = dict(update=r["update"], where_scale_min=float(res["where_scale"].min()), where_abs_max=float(res["where"].abs().max()),
                      kl_where=float(res["kl_where"]), opt_loss=float(res["opt_loss"]),
                      num_step=float(res["num_step_per_sample"].mean()),
                      grads_finite=not nonfinite_names(grads), params_finite=not nonfinite_names(p))
            trace.append(st)
            if first["loss"] is None and not np.isfinite(st["opt_loss"]):
                first["loss"] = r["update"]
            if first["grads"] is None and not st["grads_finite"]:
                first["grads"] = r["update"]
            if first["params"] is None and not st["params_finite"]:
                first["params"] = r["update"]
                break
        free[name] = dict(first_nonfinite=first, seconds=round(time.time() - t1, 1), updates_run=len(trace), trace_last=trace[-12:],
                          where_scale_min_over_window=min(t["where_scale_min"] for t in trace))
        print("oracle free run %s: first non-finite %s (engine: %d), %d updates in %.0f s" % (
            name, first, u_star, len(trace), time.time() - t1), flush=True)
    report["oracle_free_run"] = free

    # ---- 3b. teacher-forced oracle over the last updates -----------------------------------------------------------------
    k0 = max(start, u_star - args.forced)
    rec2, u2, states = replay(u_star - start, 0, want_states_from=k0)
    assert u2 == u_star, ("second replay differs from the first", u2, u_star)
    by_update = {r["update"]: r for r in rec2}
    forced = []
    for k in range(k0, u_star):
        row = dict(update=k + 1)
        after = states[k + 1] if (k + 1) in states else None
        for dtype, name in ((torch.float32, "fp32"), (torch.float64, "fp64")):
            p, slots = oracle_state(states[k], dtype)
            before = {n: v.clone() for n, v in p.items()}
            obs, noise = batch_of(by_update[k + 1], dtype)
            res, grads = O.train_step(p, slots, ocfg, obs, noise, global_step=k)
            row[name] = dict(loss_finite=bool(np.isfinite(float(res["opt_loss"]))), nonfinite_grads=nonfinite_names(grads),
                             nonfinite_params=nonfinite_names(p), where_scale_min=float(res["where_scale"].min()),
                             kl_where=float(res["kl_where"]))
            if after is not None and not row[name]["nonfinite_params"]:
                got = unflatten(after, "flat_params", torch.float64)
                if all(bool(torch.isfinite(v).all()) for v in got.values()):
                    worst = 0.0
                    for n in p:
                        d_ref = p[n].double() - before[n].double()
                        d_got = got[n] - before[n].double()
                        worst = max(worst, float((d_got - d_ref).abs().max() / (d_ref.abs().max() + 1e-30)))
                    row[name]["delta_rel_err_vs_engine"] = worst
            if k + 1 == u_star:
                eg = states["grads_at_%d" % u_star]
                report["engine_nonfinite_buffers"] = states["nonfinite_buffers_at_%d" % u_star]
                row[name]["engine_nonfinite_grads"] = nonfinite_names(eg)
                row[name]["same_nonfinite_gradient_tensors"] = nonfinite_names(eg) == nonfinite_names(grads)
                cls = lambda t: (torch.isnan(t).to(torch.int8) * 3 + torch.isposinf(t).to(torch.int8) + torch.isneginf(t).to(torch.int8) * 2)
                row[name]["elements_with_different_class"] = {
                    n: int((cls(eg[n].cpu()) != cls(grads[n].float())).sum()) for n in grads
                    if int((cls(eg[n].cpu()) != cls(grads[n].float())).sum())}
        forced.append(row)
    report["oracle_teacher_forced"] = forced
    last = forced[-1]
    report["result"] = dict(
        engine_first_nonfinite_update=u_star,
        oracle_fp32_free_run_first_nonfinite_update=free["fp32"]["first_nonfinite"]["params"],
        oracle_fp64_free_run_first_nonfinite_update=free["fp64"]["first_nonfinite"]["params"],
        oracle_fp32_from_engine_state_nonfinite_at_same_update=bool(last["fp32"]["nonfinite_params"]),
        oracle_fp64_from_engine_state_nonfinite_at_same_update=bool(last["fp64"]["nonfinite_params"]),
        same_nonfinite_gradient_tensors_fp32=last["fp32"].get("same_nonfinite_gradient_tensors"),
        worst_delta_rel_err_fp32_before=max((r["fp32"].get("delta_rel_err_vs_engine", 0.0) for r in forced[:-1]), default=None))
    path = os.path.join(args.out, "report_seed%d.json" % args.seed)
    json.dump(report, open(path, "w"), indent=1)
    print(json.dumps(report["result"], indent=1))
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
