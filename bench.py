#!/usr/bin/env python
"""bench.py -- images/sec of one AIR train step (T-step unroll forward, ELBO/NVIL backward, both RMSProp updates).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run
with one rank per GPU.  Rank 0 prints ONE JSON line.  A "step" is one pass of the hot path over one synthetic batch of
64 images per GPU (BASELINE.json configs[1]: multi-MNIST 50x50, max_steps=3, batch=64, fp32); data-parallel runs keep
64 images per GPU (weak scaling) and all-reduce the flat gradient bucket once per step over RCCL.

Besides the headline value the line carries
  roofline     : the fused ST glimpse-read kernel (north_star's target kernel): algorithmic bytes per launch
                 (4*(HW+hw+4) B per image-step) / its average duration measured here with HIP events on the
                 engine's stream, vs the 8 TB/s HBM3E peak; plus the same figure over a batch sweep (the working set
                 only leaves the 256 MiB Infinity Cache at large batch) and for the other three ST kernels.
  cpu_baseline : the CPU oracle (reference-equivalent restatement, torch-CPU fp32) timed on this host's cores on a
                 bounded sample of the same workload (kind "port").
"""
import argparse
import ctypes
import json
import os
import sys
import contextlib
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from attend_infer_repeat_amd import runtime_env as _runtime_env     # (importing the package applies the HIP runtime settings, before torch)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable copy rate



@contextlib.contextmanager
def quiet_host():
    """Timed regions run with the cyclic garbage collector off, after one explicit collection: an engine of an earlier leg that is only
    reachable through reference cycles is otherwise freed at an arbitrary allocation inside a later leg's timed loop, and freeing its
    arenas / destroying its graphs synchronises the device (the one-off +20...+40 ms stalls a 400-step region caught in about one run
    of fifteen).  Nothing of the measured step changes."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)     # 0.5 s timed: long enough to average clock / scheduling hiccups
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: 64; 1024 for --config c5)")
    ap.add_argument("--no-graph", action="store_true", help="eager C-ABI launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short captured runs of BASELINE configs[3] / configs[4] that the default headline run appends")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="time the CPU oracle on every host cpu as well (minutes on a 256-cpu host: ~100 s per step)")
    ap.add_argument("--steps-per-replay", type=int, default=1,
                    help="single GPU: consecutive updates captured per hipGraph replay (input queue of that depth; a replay "
                         "costs ~8 us on top of its nodes).  --steps is rounded up to a multiple of it")
    ap.add_argument("--breakdown", action="store_true",
                    help="time every launch of the step plan in isolation (HIP events, back-to-back repeats) -> stderr")
    ap.add_argument("--mfma", default="f32", choices=["f32", "bf16"],
                    help="dense-product precision: exact fp32 MFMA (headline) or bf16 operands / fp32 accumulate (configs[4])")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="c2: BASELINE configs[1], 50x50/20x20/T=3, batch 64, fp32 (headline); c4: configs[3], 100x100/28x28/T=5; "
                         "c5: configs[4], the c2 shapes at batch 1024 with the bf16 MFMA MLP path")
    ap.add_argument("--fixed-batch", action="store_true",
                    help="time every step on ONE fixed synthetic batch (what rounds 1-5 measured; the model collapses onto it within ~50 updates) "
                         "instead of a fresh batch per step gathered from an HBM-resident synthetic set by the first node of the captured step")
    ap.add_argument("--dataset-images", type=int, default=4096, help="size of the HBM-resident synthetic set the device feeder draws from")
    ap.add_argument("--step-bias", type=float, default=None,
                    help="probe, not a headline: bias of the steps predictor's logit (mnist_model.py:26; the script's 0.75 by default).  +20 keeps "
                         "every step present, -20 none -- measured: the step count is NOT what the state-dependent cost of the canvas kernels follows "
                         "(profiles/r05_c4_seed_dependence.txt); the line says so in config.step_bias")
    args = ap.parse_args()
    if args.config == "c5":
        args.mfma = "bf16"
    if args.batch is None:
        args.batch = 1024 if args.config == "c5" else 64
    if args.steps_per_replay > 1 or args.no_graph:
        args.fixed_batch = True          # (the input queue of a multi-step replay / the eager probe run on the observation buffers)
    return args


def event_time_ms(lib, stream_ptr, fn, reps):
    """Average duration of `fn()` (which enqueues work on the stream) measured with HIP events on that stream."""
    from attend_infer_repeat_amd import _lib
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.air_event_create(ctypes.byref(e0))); _lib.check(lib.air_event_create(ctypes.byref(e1)))
    # warm-up: three launches, then -- a fresh box clocks up over the first tens of milliseconds of load (the same 170 us launch
    # measures 192 us in its first ~50 repetitions, profiles/r04_read_placement_and_warmup.txt) -- about 40 ms of the launch itself
    for _ in range(3):
        fn()
    _lib.check(lib.air_event_record(e0, stream_ptr))
    fn()
    _lib.check(lib.air_event_record(e1, stream_ptr))
    ms1 = ctypes.c_float()
    _lib.check(lib.air_event_elapsed_ms(e0, e1, ctypes.byref(ms1)))
    # Launches of a few microseconds: a python / ctypes call per launch can be SLOWER than the kernel (a busy or slow host: the
    # same 4.0 us read measured 11-18 us per launch on two of the round's boxes), and the figure would be the host's launch rate.
    # Such launches are timed as nodes of a captured graph -- the same dependent back-to-back chain on the same stream, G launches
    # per host call.  AIR_BENCH_STREAM_TIMER=1: always the plain stream loop.  Anything going wrong falls back to it.
    if ms1.value < 0.03 and os.environ.get("AIR_BENCH_STREAM_TIMER", "0") != "1":
        G, graph, capturing = 50, ctypes.c_void_p(), False
        try:
            _lib.check(lib.air_graph_begin_capture(stream_ptr)); capturing = True
            for _ in range(G):
                fn()
            capturing = False
            _lib.check(lib.air_graph_end_capture(stream_ptr, ctypes.byref(graph)))
            for _ in range(max(2, min(400, int(40.0 / max(G * ms1.value, 1e-3))))):
                _lib.check(lib.air_graph_launch(graph, stream_ptr))
            n = max(4, reps // G * 4)
            _lib.check(lib.air_event_record(e0, stream_ptr))
            for _ in range(n):
                _lib.check(lib.air_graph_launch(graph, stream_ptr))
            _lib.check(lib.air_event_record(e1, stream_ptr))
            ms = ctypes.c_float()
            _lib.check(lib.air_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            lib.air_graph_destroy(graph)
            lib.air_event_destroy(e0); lib.air_event_destroy(e1)
            return ms.value / (n * G)
        except Exception:                      # noqa: BLE001 -- the plain loop below is always available
            if capturing:
                lib.air_graph_end_capture(stream_ptr, ctypes.byref(graph))
            if graph.value:
                lib.air_graph_destroy(graph)
    for _ in range(min(2000, int(40.0 / max(ms1.value, 1e-3)))):
        fn()
    _lib.check(lib.air_event_record(e0, stream_ptr))
    for _ in range(reps):
        fn()
    _lib.check(lib.air_event_record(e1, stream_ptr))
    ms = ctypes.c_float()
    _lib.check(lib.air_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    lib.air_event_destroy(e0); lib.air_event_destroy(e1)
    return ms.value / reps


def st_rooflines(eng, reps=200):
    """Per-launch algorithmic GB/s of the four ST kernels at the engine's own shapes and buffers."""
    import torch
    from attend_infer_repeat_amd import hip as H
    lib = H.lib()
    cfg, T, B, M = eng.cfg, eng.T, eng.B, eng.M
    (Hh, Ww), (h, w) = cfg.img_size, cfg.crop_size
    HW, hw = Hh * Ww, h * w
    p, sp = H._p, eng._sp()
    dec, dgl = eng.gd.out[-1], eng.gd.g[-1]
    # the canvas backward in the form the step runs it: the fused launch re-forms the canvas on each glimpse's footprint
    # (final_canvas = NULL) instead of reading the stored final canvas
    rc = any(e[2] == "air_canvas_unroll_fwd_bwd" for e in eng._plan_bwd)
    calls = {
        "st_read_fwd": (lambda: lib.air_st_read_fwd(p(eng.obs), p(eng.where), p(eng.glimpse_in), M, B, Hh, Ww, h, w, sp),
                        4 * (HW + hw + 4) * M),
        "st_read_bwd": (lambda: lib.air_st_read_bwd(p(eng.obs), p(eng.where), p(eng.d_glimpse_in), p(eng.dwhere_r), None,
                                                    M, B, Hh, Ww, h, w, sp), 4 * (HW + hw + 4 + 4) * M),
        "canvas_unroll_fwd": (lambda: lib.air_canvas_unroll_fwd_banded(p(dec), p(eng.where), p(eng.presence), p(eng.obs),
                                                                        p(eng.canvas_steps), p(eng.final_canvas),
                                                                        p(eng.rec_parts), eng.n_bands, T, B, Hh, Ww, h, w,
                                                                        cfg.output_multiplier, cfg.output_std, sp),
                              4 * (hw + 2 * HW + 4 + 1) * M),
        "canvas_unroll_bwd": (lambda: lib.air_canvas_unroll_bwd(p(dec), p(eng.where), p(eng.presence), p(eng.obs),
                                                                 None if rc else p(eng.final_canvas), p(dgl), p(eng.dwhere_w), T, B, Hh,
                                                                 Ww, h, w, cfg.output_multiplier, cfg.output_std,
                                                                 1.0 / B, sp), 4 * (HW + 2 * hw + 4 + 4 + 1) * M),
    }
    # the fused attend launches of the step (glimpse read + the tiny heads around it): charged with the read's bytes only
    for pname, plan, key, nb in (("attend_fwd", eng._plan_fwd_train, "air_attend_fwd", 4 * (HW + hw + 4) * M),
                                 ("attend_bwd", eng._plan_bwd, "air_attend_bwd", 4 * (HW + hw + 4 + 4) * M)):
        for fn, a, name in plan:
            if name == key:
                calls[pname] = ((lambda fn=fn, a=a: fn(*a, sp)), nb)
    # minimal bytes of the same launches: every input the launch must read once, every output it must write once (the 8(d)
    # figure charges the staged image / canvas once per glimpse)
    minimal = {
        "st_read_fwd": 4 * (B * HW + M * (hw + 4)), "attend_fwd": 4 * (B * HW + M * (hw + 4)),
        "st_read_bwd": 4 * (B * HW + M * (hw + 4 + 4)), "attend_bwd": 4 * (B * HW + M * (hw + 4 + 4)),
        "canvas_unroll_fwd": 4 * (M * (hw + 5) + B * HW * (2 + (T if eng.canvas_steps is not None else 0))),
        # stored-canvas form: final canvas + obs once per image; recompute form: obs + the image's T glimpses per (t, b) unit
        # are re-read from L2, the bytes that MUST move are obs, glimpse, dglimpse, where, dwhere, presence
        "canvas_unroll_bwd": 4 * ((1 if rc else 2) * B * HW + M * (2 * hw + 9)),
    }
    pmc, pmc_note = pmc_traffic(lib, (Hh, Ww, h, w, T, B))
    out = {}
    for name, (fn, nbytes) in calls.items():
        ms = event_time_ms(lib, sp, fn, reps)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": pmc.get(name, {}).get("traffic_bytes"),
                     "us_per_launch": round(ms * 1e3, 3), "algorithmic_bytes_per_launch": nbytes,
                     "minimal_bytes_per_launch": minimal[name],
                     "frac_minimal_bytes": round(minimal[name] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        if pmc_note:
            out[name]["traffic_note"] = pmc_note
    out["canvas_unroll_bwd"]["form"] = "recompute (no final-canvas read)" if rc else "stored final canvas"
    return out


def pmc_traffic(lib, shape):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (tools/profile_round.sh -> tools/pmc_to_json.py).
    The file records the digest of the sources the profiled binary was built from; a file from another binary or another
    shape is NOT used (traffic = null) instead of silently going stale."""
    import glob
    digest = lib.air_build_digest().decode()
    best, note = {}, "no profiles/*_instep_pmc.json for this binary (digest %s) and shape" % digest[:12]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_instep_pmc.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("build_digest") == digest and tuple(d.get("shape", ())) == tuple(shape):
            best, note = d["kernels"], None
    return best, note


def instep_durations(lib, shape, mfma_dtype, n_launches=None):
    """In-graph average duration of every kernel of the replayed step from the committed rocprofv3 kernel trace of THIS binary,
    shape and PLAN (tools/profile_round.sh -> tools/rocpd_summary.py --json): {kernel-name prefix: avg_us}, or {} when no file matches.
    The profiles are taken on the default plan: with any plan-changing AIR_* switch set, or when the step's launch count differs from
    the profiled one, nothing is returned (a duration measured on a different plan must not sit next to live numbers)."""
    import glob
    if plan_env_overrides():
        return {}, None
    digest = lib.air_build_digest().decode()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_instep_durations.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("build_digest") == digest and tuple(d.get("shape", ())) == tuple(shape) and d.get("mfma_dtype", "f32") == mfma_dtype:
            if n_launches is not None and len(d["positions"]) != n_launches:
                continue
            out = {}
            for e in d["positions"]:
                k = e["kernel"].split("<")[0]
                out.setdefault(k, []).append(e["avg_us"])
            return {k: sum(v) / len(v) for k, v in out.items()}, os.path.basename(path)
    return {}, None


PLAN_ENV = ("AIR_DEFER_DW_MIN_ROWS", "AIR_FUSE_ATTEND_M", "AIR_FUSE_LSTM_TILES", "AIR_FUSE_LSTM_WIDE", "AIR_SPLIT_K0", "AIR_OPT_RIDERS",
            "AIR_FUSE_CANVAS", "AIR_FUSE_CANVAS_THROUGHPUT", "AIR_CANVAS_SPLIT", "AIR_BF16_STORAGE", "AIR_BF16_LSTM", "AIR_OPT_FOLD",
            "AIR_GEMM_WIDE_MIN_TILES", "AIR_GEMM_WIDE_TN_BF16", "AIR_GEMM_WIDE_TN_F32", "AIR_GEMM_WIDE_NT_K", "AIR_GEMM_BIG_XCD",
            "AIR_GEMM_BF16_STORAGE", "AIR_LSTM_DW_EARLY", "AIR_FUSE_WHAT_HEAD", "AIR_FUSE_GAUSS_BWD", "AIR_FOLD_GX", "AIR_FUSE_PROLOGUE_CVT")


def plan_env_overrides():
    """the plan-changing developer switches set in this process: a committed in-graph profile was taken without any"""
    return {k: os.environ[k] for k in PLAN_ENV if k in os.environ}


def in_step_event_us(eng, key, steps=200, warm=30):
    """Duration of the plan entry `key` INSIDE the train step, live: the step's own launch list is issued eagerly on the engine
    stream (same kernels, same order, same buffers as the captured graph) with a HIP event recorded right before and right after
    that one entry, so the kernel finds its operands exactly as cold as it does in the replayed step (its inputs were written by the
    launch in front of it; the weights were last touched a step ago).  Average over `steps` steps.  The interval runs from the end of
    the previous launch to the end of this one, i.e. it includes this entry's dispatch gap: an upper bound of the kernel's own
    duration, which the committed rocprofv3 positions file gives (`in_graph_profiled`)."""
    from attend_infer_repeat_amd import hip as H, _lib
    lib, sp = H.lib(), eng._sp()
    plans = eng._single_gpu_step_plans() if eng.world_size == 1 else [eng._plan_fwd_train, eng._plan_bwd, eng._plan_opt]
    flat = [e for pl in plans for e in pl]
    idx = [i for i, e in enumerate(flat) if e[2] == key]
    if not idx:
        return None
    i0 = idx[0]
    e0s = [ctypes.c_void_p() for _ in range(steps)]
    e1s = [ctypes.c_void_p() for _ in range(steps)]
    for e in e0s + e1s:
        _lib.check(lib.air_event_create(ctypes.byref(e)))
    for it in range(warm + steps):
        k = it - warm
        for j, e in enumerate(flat):
            if j == i0 and k >= 0:
                _lib.check(lib.air_event_record(e0s[k], sp))
            st = e[0](*e[1], sp)
            if st != 0:
                _lib.check(st, e[2])
            if j == i0 and k >= 0:
                _lib.check(lib.air_event_record(e1s[k], sp))
    eng.synchronize()
    tot = 0.0
    for a, b in zip(e0s, e1s):
        ms = ctypes.c_float()
        _lib.check(lib.air_event_elapsed_ms(a, b, ctypes.byref(ms)))
        tot += ms.value
    for e in e0s + e1s:
        lib.air_event_destroy(e)
    return tot / steps * 1e3


def attend_roofline(eng, lib, live_warm_us=None):
    """`roofline` of the fused glimpse read as it runs in the step (SURVEY 8(d) read bytes / duration).  Primary figure: the in-step
    duration -- from the committed rocprofv3 positions of THIS binary, shape and plan when there is one, else the live in-step event
    timing; the back-to-back (warm-operand) launch time is kept as `frac_live`."""
    cfg, T, B, M = eng.cfg, eng.T, eng.B, eng.M
    (Hh, Ww), (h, w) = cfg.img_size, cfg.crop_size
    nbytes = 4 * (Hh * Ww + h * w + 4) * M
    us_ev = in_step_event_us(eng, "air_attend_fwd")
    if us_ev is None:
        return None
    n_launch = sum(eng.kernel_launch_count().values())
    dur, dur_file = instep_durations(lib, (Hh, Ww, h, w, T, B), cfg.mfma_dtype, n_launch)
    us_prof = dur.get("attend_fwd_kernel")
    us = us_prof if us_prof else us_ev
    gbs = lambda u: nbytes / (u * 1e-6) / 1e9
    out = {"bound": "hbm", "kernel": "attend_fwd_kernel (fused glimpse read of all T steps + where sampling + presence / num-steps heads)",
           "achieved": round(gbs(us), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs(us) / HBM_PEAK_GBS, 5),
           "us_per_launch": round(us, 3), "algorithmic_bytes_per_launch": nbytes,
           "duration_source": ("profiles/%s (rocprofv3 kernel trace of the replayed graph, this binary / shape / plan)" % dur_file) if us_prof
                              else "live: HIP events around the entry inside the eagerly issued step (includes its dispatch gap)",
           "in_step_events": {"us_per_launch": round(us_ev, 3), "frac": round(gbs(us_ev) / HBM_PEAK_GBS, 5)}}
    if us_prof:
        out["in_graph_profiled"] = {"us_per_launch": round(us_prof, 3), "frac": round(gbs(us_prof) / HBM_PEAK_GBS, 5), "source": "profiles/" + dur_file}
    if live_warm_us:
        out["frac_live"] = round(gbs(live_warm_us) / HBM_PEAK_GBS, 5)
        out["us_per_launch_live_back_to_back"] = round(live_warm_us, 3)
    return out


FEEDER_IMAGES = 4096


def resident_dataset(img_size, max_objects, device, n=FEEDER_IMAGES, seed=0, _cache={}):
    """HBM-resident synthetic set for the device feeder (attend_infer_repeat_amd.engine.AIREngine.attach_dataset: every captured step
    starts with air_batch_gather drawing its own batch with replacement, the reference's data.py:121-158 feeder moved into HBM)."""
    import torch
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    key = (tuple(img_size), max_objects, n, seed, str(device))
    if key not in _cache:
        _cache.clear()                                        # (one set at a time: the 100x100 one is 164 MB)
        imgs, _ = synthetic_multi_mnist(n, img_size, max_objects=max_objects, seed=seed)
        _cache[key] = torch.from_numpy(imgs).reshape(n, -1).contiguous().to(device)
    return _cache[key]


def run_other_config(name, device, steps=400, warmup=100, seed=1, feeder=True, step_bias=None):
    """A short captured run of another named single-GPU configuration (BASELINE configs[3] = c4, configs[4] = c5) inside the default
    invocation, so that the driver's line carries driver-observed numbers for every single-GPU configuration."""
    import torch
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    from attend_infer_repeat_amd import hip as H
    kw = dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5) if name == "c4" else {}
    B = 1024 if name == "c5" else 64
    if step_bias is not None:
        kw = dict(kw, step_bias=float(step_bias))
    cfg = EngineConfig(mfma_dtype="bf16" if name == "c5" else "f32", **kw)
    eng = AIREngine(cfg, B, device=device, seed=seed, keep_canvas_steps=True)
    imgs, _ = synthetic_multi_mnist(B, cfg.img_size, max_objects=4 if name == "c4" else 2, seed=0)
    eng.set_obs(torch.from_numpy(imgs).to(device))
    if feeder:
        eng.attach_dataset(resident_dataset(cfg.img_size, 4 if name == "c4" else 2, device), shuffle=True, seed=seed)
    eng.capture()
    for _ in range(warmup):
        eng.train_step()
    torch.cuda.synchronize(device)
    with quiet_host():
        rec_t = _timed_blocks(eng, steps, device)
    blocks, t_blocks, el = rec_t
    return _finish_other_config(name, eng, cfg, B, steps, warmup, blocks, t_blocks, el, feeder)


def _timed_blocks(eng, steps, device):
    import torch
    # timed as `blocks` consecutive blocks (a synchronize between them: ~10 us each): `ms_per_step` is the WHOLE region, the median block
    # rides beside it -- a 400-step region of 130 ms caught a one-off +20 ms stall in about one run of fifteen (0.3577 against 0.3021-0.3030)
    blocks, t_blocks = 4 if steps % 4 == 0 and steps >= 40 else 1, []
    t0 = time.perf_counter()
    for _ in range(blocks):
        tb = time.perf_counter()
        for _ in range(steps // blocks):
            eng.train_step()
        torch.cuda.synchronize(device)
        t_blocks.append((time.perf_counter() - tb) / (steps // blocks) * 1e3)
    el = time.perf_counter() - t0
    return blocks, t_blocks, el


def _finish_other_config(name, eng, cfg, B, steps, warmup, blocks, t_blocks, el, feeder):
    import gc
    import torch
    from attend_infer_repeat_amd import hip as H
    finite = bool(torch.isfinite(eng.flat_params).all().item())
    # the model state the last timed step ran in (what the data-dependent canvas kernels saw): objects per image, |where| per component
    eng.synchronize()
    state = {"steps_present_per_image": round(float(eng.presence.sum(0).mean().item()), 3),
             "mean_abs_where": [round(float(v), 3) for v in eng.where.abs().mean(dim=(0, 1)).tolist()]}
    eng.release_graphs()
    roof = attend_roofline(eng, H.lib())
    (Hh, Ww), (hh, ww) = cfg.img_size, cfg.crop_size
    rec = {"workload": f"{Hh}x{Ww} canvas, max_steps={eng.T}, glimpse {hh}x{ww}, batch={B}, "
                       f"{'bf16-operand MFMA MLP path' if name == 'c5' else 'fp32'}, hipGraph replay "
                       f"(BASELINE configs[{3 if name == 'c4' else 4}])",
           "value": round(B * steps / el, 1), "unit": "images/sec", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
           "warmup": warmup, "median_block_ms_per_step": round(sorted(t_blocks)[len(t_blocks) // 2], 4), "blocks": blocks,
           "kernel_launches_per_step": sum(eng.kernel_launch_count().values()),
           "input": ("fresh batch per step: air_batch_gather from %d HBM-resident synthetic images, first node of the captured step" % FEEDER_IMAGES)
                    if feeder else "one fixed synthetic batch",
           "step_bias": cfg.step_bias,
           "params_finite_after_run": finite, "model_state_at_end": state, "roofline": roof}
    if feeder:
        eng.attach_dataset(None)
    del eng
    gc.collect()                # (the engine is freed HERE, not inside the next leg's timed loop: see quiet_host)
    torch.cuda.empty_cache()
    return rec


C4_SEEDS = (1, 1000004, 7)      # 1000004 = distributed.rank_seed(1, 0), the seed of `bench.py --config c4`


def run_other_config_seeds(name, device):
    """configs[3]'s step time follows the MODEL STATE, not only the shapes: the canvas-write backward costs what the glimpses it
    writes cover (18-39 us in one and the same step position, profiles/r05_c4_seed_dependence.txt), and a few hundred updates from
    a random initialisation take different engine seeds (initial parameters + noise stream) to different `where` distributions --
    0.298-0.333 ms over four seeds on one box and binary.  The line therefore carries configs[3] as the aggregate of C4_SEEDS
    (total images / total time) with every seed's own figure beside it; batch 1024 (c5) averages over sixteen times the images
    and stays on one seed."""
    def beside(rec, **kw):
        r = run_other_config(name, device, steps=200, warmup=100, **kw)
        return {k: r[k] for k in ("ms_per_step", "value", "steps", "warmup", "input", "step_bias", "kernel_launches_per_step", "model_state_at_end")}
    if name != "c4":
        rec = run_other_config(name, device)
        rec["fixed_batch"] = beside(rec, feeder=False)                  # rounds 1-5's input, for continuity
        rec["all_steps_present_probe"] = beside(rec, step_bias=20.0)    # every step present (the image-major canvas backward skips absent ones)
        return rec
    recs = [run_other_config(name, device, seed=s) for s in C4_SEEDS]
    rec = dict(recs[0])
    total_ms = sum(r["ms_per_step"] * r["steps"] for r in recs)
    total_steps = sum(r["steps"] for r in recs)
    rec["ms_per_step"] = round(total_ms / total_steps, 4)
    rec["value"] = round(64 * total_steps / (total_ms * 1e-3), 1)
    rec["steps"] = total_steps
    rec["params_finite_after_run"] = all(r["params_finite_after_run"] for r in recs)
    rec["median_block_ms_per_step"] = round(sum(r.get("median_block_ms_per_step", r["ms_per_step"]) for r in recs) / len(recs), 4)
    rec["per_seed"] = [{"engine_seed": s, "ms_per_step": r["ms_per_step"], "median_block_ms_per_step": r.get("median_block_ms_per_step"),
                        "value": r["value"], "model_state_at_end": r["model_state_at_end"]}
                       for s, r in zip(C4_SEEDS, recs)]
    rec["note"] = ("aggregate over %d engine seeds x %d timed steps (rounds 2-5: the step time of this configuration followed the model "
                   "state through the canvas-write backward, 0.298-0.333 ms; round 6's glimpse-space backward: see per_seed)" % (len(recs), recs[0]["steps"]))
    ms = [r.get("median_block_ms_per_step") or r["ms_per_step"] for r in recs]    # (median block: a one-off stall is not state dependence)
    rec["per_seed_spread"] = round((max(ms) - min(ms)) / min(ms), 4)
    rec["fixed_batch"] = beside(rec, feeder=False)
    rec.update(c4_shape_sweeps(device))
    return rec


def c4_shape_sweeps(device, pts=(1024, 8192, 65536)):
    """The ST bandwidth study at configs[3]'s shapes (BASELINE configs[3] "bandwidth-bound ST kernel") on the default line
    (VERDICT r05 item 2a): out of cache from 8192 images on (100x100 fp32 = 40 KB per image)."""
    from attend_infer_repeat_amd.engine import EngineConfig
    cfg4 = EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5)
    out = {"roofline_sweep_st_read_fwd": st_read_sweep(cfg4, 5, list(pts), device)}
    cw_f, cw_b, cw_i = canvas_write_sweep(cfg4, 5, list(pts), device)
    out["roofline_sweep_canvas_write_fwd"], out["roofline_sweep_canvas_write_bwd"], out["roofline_sweep_canvas_write_pair"] = cw_f, cw_b, cw_i
    return out


# DESIGN section 5's prediction for first contact with a multi-GPU node, kept NEXT TO the code that prints the measurement (VERDICT r05
# item 4c): weak-scaling efficiency at 8 ranks (64 images per GPU, 0.20 ms single-GPU step, 10.5 MB gradient bucket) per protocol, from
# a 30-50 us latency floor + 2(N-1)/N x bytes at 100-200 GB/s of achieved bus bandwidth for a library all-reduce of this size, and
# 5-10 us per barrier + 1.3 MB per xGMI link at 60-120 GB/s each way for the direct exchange.  The ORDERING is the claim.
PREDICTED_AT_8_RANKS = {
    "torch-split": {"step_ms": [0.32, 0.43], "efficiency": [0.47, 0.63], "exposed": "the whole 10.5 MB all-reduce behind the backward"},
    "torch-overlap": {"step_ms": [0.31, 0.44], "efficiency": [0.46, 0.65], "exposed": "4.8 MB tail partly under the backward (45 us hide), 5.7 MB head fully"},
    "rccl-captured": {"step_ms": [0.29, 0.42], "efficiency": [0.48, 0.69], "exposed": "as above minus the host round trips"},
    "ipc-rsag": {"step_ms": [0.22, 0.25], "efficiency": [0.81, 0.92], "exposed": "2 barriers + shard pull + shard push over 7 links at once, replacing the closing update"},
}


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X fp32 matrix peak (v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate)


def gemm_roofline(eng, reps=50):
    """MFMA utilisation of the dense layers: 2*M*N*K summed over every GEMM problem of the step / the summed isolated
    launch time of those launches (HIP events).  At batch 64 this is a latency statement, not a throughput one."""
    from attend_infer_repeat_amd import hip as H
    lib = H.lib()
    sp = eng._sp()
    flops, us, launches = 0.0, 0.0, 0
    for plan in (eng._plan_fwd_train, eng._plan_bwd):
        for fn, a, name in plan:
            if name in ("air_gemm", "air_gemm_bf16"):
                f = 2.0 * a[2] * a[3] * a[4]
            elif name in ("air_gemm_grouped", "air_gemm_grouped_gather"):
                f = sum(2.0 * a[0][q].M * a[0][q].N * a[0][q].K for q in range(a[1]))
            elif name in ("air_lstm_step_fwd", "air_lstm_step_fwd_prologue"):   # h[M,Hd] . W_h[Hd,4Hd], gate math fused
                f = 2.0 * a[9] * a[10] * 4 * a[10]
            elif name == "air_what_head_fwd":                      # ge_out[T*B,K] . W[K,2A], sampling fused
                f = 2.0 * a[20] * a[21] * 2 * a[14] * a[2]
            elif name == "air_lstm_first_step_fwd":                # x[M,E] . W_x[E,4Hd] + h0 . W_h, gate math fused
                f = 2.0 * a[14] * 4 * a[15] * (a[2] + a[15])
            elif name == "air_lstm_step_bwd":                      # dgates[M,4Hd] . W_h^T
                f = 2.0 * a[12] * a[13] * 4 * a[13]
            else:
                continue
            flops += f
            us += event_time_ms(lib, sp, lambda: fn(*a, sp), reps) * 1e3
            launches += 1
    tf = flops / (us * 1e-6) / 1e12
    # bf16 mode: v_mfma_f32_16x16x16_bf16, half the K per instruction of the gfx950 16x16x32 form -> price against the
    # dense bf16 peak of MI355X_MICROARCH.md (2.5 PF) all the same; the operands are still fetched as fp32
    peak = F32_MFMA_PEAK_TFLOPS if eng.cfg.mfma_dtype == "f32" else 2516.6
    out = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s",
           "frac": round(tf / peak, 5), "traffic": None, "gemm_launches": launches,
           "gemm_flops_per_step": int(flops), "gemm_us_per_step_isolated": round(us, 1)}
    # MFMA utilisation from the counters (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES and against the dispatches' wall time) of the
    # committed PMC pass of THIS binary and shape (tools/profile_round.sh -> tools/rocpd_pmc.py --mfma-json), next to FLOP/s / peak
    import glob
    (Hh, Ww), (h, w) = eng.cfg.img_size, eng.cfg.crop_size
    digest = lib.air_build_digest().decode()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma_util.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("build_digest") == digest and d.get("shape") == [Hh, Ww, h, w, eng.T, eng.B] and d.get("mfma_dtype", "f32") == eng.cfg.mfma_dtype:
            out["mfma_busy_utilisation"] = dict(d["dense_kernels_total"], source="profiles/" + os.path.basename(path))
            break
    return out


def plan_breakdown(eng, reps=100):
    """Per-launch steady-state cost (launch + execution) of each entry of the step plan, run back-to-back in isolation."""
    from attend_infer_repeat_amd import hip as H
    lib = H.lib()
    sp = eng._sp()
    rows = []
    for phase, plan in (("fwd", eng._plan_fwd_train), ("bwd", eng._plan_bwd), ("opt", eng._plan_opt)):
        for i, (fn, a, name) in enumerate(plan):
            ms = event_time_ms(lib, sp, lambda: fn(*a, sp), reps)
            desc = ""
            if name in ("air_gemm", "air_gemm_bf16"):
                desc = f"ta={a[0]} tb={a[1]} M={a[2]} N={a[3]} K={a[4]} epi={a[12]}"
            elif name in ("air_gemm_grouped", "air_gemm_grouped_gather"):
                desc = " | ".join(f"{'T' if d.ta else 'N'}{'T' if d.tb else 'N'} {d.M}x{d.N}x{d.K}" for d in a[0])
            rows.append((phase, i, name, desc, ms * 1e3))
    tot = sum(r[4] for r in rows)
    print(f"# isolated per-launch cost, {len(rows)} launches, sum {tot:.1f} us", file=sys.stderr)
    for r in rows:
        print(f"{r[0]:4s} {r[1]:3d} {r[2]:28s} {r[4]:8.2f} us  {r[3]}", file=sys.stderr)
    agg = {}
    for r in rows:
        agg.setdefault(r[2], [0, 0.0]); agg[r[2]][0] += 1; agg[r[2]][1] += r[4]
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"## {k:28s} x{n:3d} {t:8.1f} us ({100 * t / tot:4.1f}%)", file=sys.stderr)


def st_read_sweep(cfg, T, batches, device, share_image=True):
    """ST glimpse read over a batch sweep: the >=70%-of-roofline target is only meaningful once the working set
    (B*10 KB images + T*B*1.6 KB glimpses) exceeds the 256 MiB Infinity Cache."""
    import torch
    from attend_infer_repeat_amd import hip as H
    lib = H.lib()
    (Hh, Ww), (h, w) = cfg.img_size, cfg.crop_size
    HW, hw = Hh * Ww, h * w
    res = []
    stream = torch.cuda.Stream(device=device)
    sp = ctypes.c_void_p(stream.cuda_stream)
    for B in batches:
        n = T * B
        n_img = B if share_image else n
        img = torch.rand(n_img, Hh, Ww, device=device)
        where = torch.empty(n, 4, device=device)
        where[:, 0] = 0.45 + 0.2 * torch.rand(n, device=device); where[:, 2] = 0.45 + 0.2 * torch.rand(n, device=device)
        where[:, 1] = 0.6 * torch.rand(n, device=device) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=device) - 0.3
        out = torch.empty(n, h, w, device=device)
        torch.cuda.synchronize()
        fn = lambda: lib.air_st_read_fwd(H._p(img), H._p(where), H._p(out), n, n_img, Hh, Ww, h, w, sp)
        ms = event_time_ms(lib, sp, fn, 20 if B >= 16384 else 100)
        nbytes = 4 * (HW + hw + 4) * n                      # the image charged once per glimpse
        minimal = 4 * (n_img * HW + n * (hw + 4))            # what the launch must move: each image once, each glimpse once
        gbs, gbs_min = nbytes / (ms * 1e-3) / 1e9, minimal / (ms * 1e-3) / 1e9
        # (`frac` from the bytes the launch must move, so it cannot exceed 1; the SURVEY 8(d) figure -- image charged once per
        #  glimpse -- is kept as a rate only: with T glimpses per staged image it over-counts the traffic by design)
        res.append({"batch": B, "glimpses": n, "images": n_img, "us_per_launch": round(ms * 1e3, 2),
                    "working_set_MiB": round((n_img * HW + n * hw) * 4 / 2 ** 20, 1),
                    "achieved_minimal_bytes": round(gbs_min, 1), "frac": round(gbs_min / HBM_PEAK_GBS, 4),
                    "achieved_survey_8d_bytes": round(gbs, 1)})
        del img, where, out
    return res


def canvas_write_sweep(cfg, T, batches, device, scale=(0.45, 0.2)):
    """The inverse canvas write (north_star names it next to the read) over a batch sweep, forward (T inverse warps + canvas
    accumulation + reconstruction term, per-step canvases kept as the API path keeps them) and backward (dcanvas formed on the
    fly -> dglimpse, dwhere).  Fractions from the bytes each launch must move once; SURVEY 8(d)'s per-(image, step) figures
    (21,620 / 13,236 B at 50x50 / 20x20) next to them as rates."""
    import torch
    from attend_infer_repeat_amd import hip as H
    lib = H.lib()
    (Hh, Ww), (h, w) = cfg.img_size, cfg.crop_size
    HW, hw = Hh * Ww, h * w
    stream = torch.cuda.Stream(device=device)
    sp = ctypes.c_void_p(stream.cuda_stream)
    fwd, bwd, fused = [], [], []
    for B in batches:
        n = T * B
        g = torch.Generator(device=device).manual_seed(B)
        glm = torch.randn(n, hw, device=device, generator=g)
        where = torch.empty(n, 4, device=device)
        where[:, 0] = scale[0] + scale[1] * torch.rand(n, device=device, generator=g); where[:, 2] = scale[0] + scale[1] * torch.rand(n, device=device, generator=g)
        where[:, 1] = 0.6 * torch.rand(n, device=device, generator=g) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=device, generator=g) - 0.3
        pres = (torch.rand(n, device=device, generator=g) < 0.7).float()
        obs = torch.rand(B, HW, device=device, generator=g)
        steps = torch.empty(T, B, HW, device=device)
        final = torch.empty(B, HW, device=device)
        nb = int(lib.air_canvas_unroll_bands(B, Hh))
        parts = torch.empty(nb, B, device=device)
        dgl = torch.empty(n, hw, device=device)
        dwh = torch.empty(n, 4, device=device)
        torch.cuda.synchronize()
        p = H._p
        f = lambda: lib.air_canvas_unroll_fwd_banded(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, T, B,
                                                     Hh, Ww, h, w, cfg.output_multiplier, cfg.output_std, sp)
        b = lambda: lib.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), p(final), p(dgl), p(dwh), T, B, Hh, Ww, h, w,
                                              cfg.output_multiplier, cfg.output_std, 1.0 / B, sp)
        reps = 10 if B >= 16384 else 100
        cases = [("fwd", f, fwd, 4 * (n * (hw + 5) + B * HW * (2 + T)), 4 * (hw + 2 * HW + 5) * n),
                 ("bwd", b, bwd, 4 * (2 * B * HW + n * (2 * hw + 9)), 4 * (HW + 2 * hw + 9) * n)]
        for name, fn, out, minimal, survey in cases:
            ms = event_time_ms(lib, sp, fn, reps)
            gmin = minimal / (ms * 1e-3) / 1e9
            out.append({"batch": B, "glimpses": n, "us_per_launch": round(ms * 1e3, 2), "row_bands": nb,
                        "working_set_MiB": round(minimal / 2 ** 20, 1), "minimal_bytes_per_launch": minimal,
                        "achieved_minimal_bytes": round(gmin, 1), "frac": round(gmin / HBM_PEAK_GBS, 4),
                        "achieved_survey_8d_bytes": round(survey / (ms * 1e-3) / 1e9, 1)})
        del glm, where, pres, obs, steps, final, parts, dgl, dwh
    for f_, b_ in zip(fwd, bwd):        # the pair, forward + backward launches back to back, on the bytes both must move once
        us = f_["us_per_launch"] + b_["us_per_launch"]
        mb = f_["minimal_bytes_per_launch"] + b_["minimal_bytes_per_launch"]
        fused.append({"batch": f_["batch"], "us_fwd_plus_bwd": round(us, 2), "minimal_bytes": mb,
                      "achieved_minimal_bytes": round(mb / (us * 1e-6) / 1e9, 1), "frac": round(mb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
    return fwd, bwd, fused


def stream_reference(device, mib=1024):
    """What this box sustains for a plain device-to-device copy of a working set far beyond the 256 MiB Infinity Cache
    (torch's copy kernel: plumbing, only used as the yardstick next to the 8 TB/s spec peak)."""
    import torch
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(80):                      # ~40 ms: the same warm-up the event-timed kernels get (event_time_ms)
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(20):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / 20
    gbs = 2 * n * 4 / (ms * 1e-3) / 1e9
    del a, b
    return {"kind": "torch device-to-device copy, 1 GiB read + 1 GiB written", "achieved": round(gbs, 1), "unit": "GB/s",
            "frac_of_spec_peak": round(gbs / HBM_PEAK_GBS, 4)}


def cpu_baseline(cfg_kw, batch, seconds, all_cores_flag=False):
    """The CPU oracle's full train step (reference-equivalent restatement) on this host's cores, bounded sample."""
    import torch
    from oracle import air_oracle as O
    ocfg = O.AIRConfig(**cfg_kw)
    params = O.init_params(ocfg, seed=1)
    slots = O.rmsprop_init(params)
    obs, _ = O.synthetic_batch(ocfg, batch, seed=0)
    # pick the thread count that serves this small-op workload best on this host (more threads != faster here)
    avail = os.cpu_count() or 1
    best = (None, 1e9)
    for th in sorted({4, 8, 16, 32}):
        if th > avail:
            continue
        torch.set_num_threads(th)
        O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=90), global_step=0)
        t0 = time.perf_counter()
        for i in range(2):
            O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=91 + i), global_step=0)
        dt = (time.perf_counter() - t0) / 2
        if dt < best[1]:
            best = (th, dt)
    cores = best[0]
    # per-core figure: the same step on ONE thread, a few steps only
    torch.set_num_threads(1)
    O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=80), global_step=0)
    t0, n1 = time.perf_counter(), 0
    while n1 < 20 and (n1 < 3 or time.perf_counter() - t0 < 3.0):
        O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=81 + n1), global_step=0)
        n1 += 1
    one_thread = batch * n1 / (time.perf_counter() - t0)
    # ... and on ALL host cores (SURVEY 8d asks for it).  On these small ops more threads are SLOWER, dramatically so on a
    # 256-cpu host: 0.6 images/s measured with 256 threads (profiles/r03_a_bench_c2_b64_all_cores.json: ~100 s per step), which
    # would blow the few-minutes budget of a default run -- so the all-cores leg runs up to 64 threads by default and the
    # full-host figure only with --cpu-all-cores
    all_threads = avail if (all_cores_flag or avail <= 64) else 64
    torch.set_num_threads(all_threads)
    t0 = time.perf_counter()
    O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=70), global_step=0)
    first = time.perf_counter() - t0
    na, t0 = 0, time.perf_counter()
    while first < 5.0 and na < 10 and (na < 1 or time.perf_counter() - t0 < 3.0):
        O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=71 + na), global_step=0)
        na += 1
    all_cores = batch * na / (time.perf_counter() - t0) if na else batch / first
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=200 + n), global_step=2 + n)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 400:
            break
    return {"value": round(batch * n / el, 1), "unit": "images/sec", "cores": cores, "kind": "port",
            "one_thread_value": round(one_thread, 1),
            # the "N = all host cores" leg of SURVEY 8(d): run on `many_threads` threads -- ALL host cpus only when that is <= 64 or
            # --cpu-all-cores was given (256 threads on these small ops take ~100 s per step: profiles/r03_a_*_all_cores.json)
            "many_threads_value": round(all_cores, 2), "many_threads": all_threads, "host_cpus": avail,
            "all_host_cores_value": round(all_cores, 2) if all_threads == avail else None,
            # ... and all host cores the way a CPU would be used for this workload: independent replicas, rates added
            "all_host_cores_replicas": cpu_all_cores_replicas(cfg_kw, batch, cores),
            "sample": f"{n} full train steps at batch {batch} ({el:.1f} s) of the torch-CPU fp32 oracle "
                      f"(oracle/air_oracle.py); threads={cores} chosen as the fastest of a sweep on this {avail}-cpu host"}


_REPLICA_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
from oracle import air_oracle as O
cfg_kw, batch, threads, seconds, seed = json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
torch.set_num_threads(threads)
ocfg = O.AIRConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in cfg_kw.items()})
params = O.init_params(ocfg, seed=1); slots = O.rmsprop_init(params)
obs, _ = O.synthetic_batch(ocfg, batch, seed=seed)
O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=90), global_step=0)
print("READY", flush=True)
sys.stdin.readline()                       # every replica starts its timed steps on the parent's signal
n, t0 = 0, time.perf_counter()
while n < 2 or time.perf_counter() - t0 < seconds:
    O.train_step(params, slots, ocfg, obs, O.make_noise(ocfg, batch, seed=200 + n), global_step=2 + n)
    n += 1
print("RATE %.3f %d" % (batch * n / (time.perf_counter() - t0), n), flush=True)
"""


def cpu_all_cores_replicas(cfg_kw, batch, threads_per_replica, seconds=4.0, limit_s=90.0):
    """SURVEY 8(d)'s "N = all host cores" as the CPU would be used for this workload: host_cpus / threads_per_replica INDEPENDENT
    replicas of the oracle's train step (fresh interpreters, one batch each -- the data-parallel shape the multi-GPU line has),
    started together, their rates added.  (ONE step on all cores is pathological for these small ops: 0.6 images/s on 256 threads.)
    Returns None when anything goes wrong or takes longer than limit_s: the leg never blocks the line."""
    import subprocess
    avail = os.cpu_count() or 1
    n_rep = max(1, avail // max(1, threads_per_replica))
    root = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_replica), MKL_NUM_THREADS=str(threads_per_replica), HIP_VISIBLE_DEVICES="")
    procs = []
    t_start = time.perf_counter()
    try:
        for r in range(n_rep):
            procs.append(subprocess.Popen([sys.executable, "-c", _REPLICA_WORKER, root, json.dumps(cfg_kw), str(batch),
                                           str(threads_per_replica), str(seconds), str(r)], stdin=subprocess.PIPE,
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env))
        import select
        for pr in procs:                       # wait until every replica has imported torch and run its warm-up step
            left = limit_s - (time.perf_counter() - t_start)
            if left <= 0 or not select.select([pr.stdout], [], [], left)[0] or not pr.stdout.readline().startswith("READY"):
                raise RuntimeError("a replica did not come up")
        for pr in procs:
            pr.stdin.write("go\n"); pr.stdin.flush()
        total, steps = 0.0, 0
        for pr in procs:
            left = max(1.0, limit_s - (time.perf_counter() - t_start))
            if not select.select([pr.stdout], [], [], left)[0]:
                raise RuntimeError("a replica did not finish")
            tok = pr.stdout.readline().split()
            if len(tok) != 3 or tok[0] != "RATE":
                raise RuntimeError("a replica failed")
            total += float(tok[1]); steps += int(tok[2])
        return {"value": round(total, 1), "replicas": n_rep, "threads_per_replica": threads_per_replica,
                "cores": n_rep * threads_per_replica, "host_cpus": avail,
                "sample": f"{steps} train steps at batch {batch} over {n_rep} independent replicas ({seconds:.0f} s each, started together)"}
    except Exception:                          # noqa: BLE001 -- a baseline leg must never cost the line
        return None
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=5)
            except Exception:                  # noqa: BLE001
                pass


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with one rank per GPU on this node
    (exactly what the driver's own command line does), so that a plain invocation IS an N-rank run."""
    import socket
    import subprocess
    from attend_infer_repeat_amd.distributed import free_rendezvous_port
    port = free_rendezvous_port()            # (not bind(0): an ephemeral port can be taken by a waiting rank's own connect attempt)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    import torch
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # AIR_BENCH_SHARE_GPU=1 (testing aid, tests/test_bench_multirank.py): every rank on GPU 0 with the gloo backend, so that the
    # N > 1 control flow of this script -- barriers, the max-over-ranks clock, collectives inside the timed steps -- can be
    # exercised on a single-GPU box.  Never a measurement.
    share_gpu = os.environ.get("AIR_BENCH_SHARE_GPU", "0") == "1"
    if torch.cuda.device_count() < args.gpus and not share_gpu:
        raise SystemExit(f"--gpus {args.gpus} but this node exposes {torch.cuda.device_count()} GPU(s)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from attend_infer_repeat_amd import build as air_build
    air_build.build()
    import torch.distributed as dist
    from attend_infer_repeat_amd import distributed as D
    if world > 1:
        D.init_from_env(backend="gloo" if share_gpu else "nccl")   # "nccl" = RCCL over xGMI
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    from attend_infer_repeat_amd import hip as H, _lib

    cfg_kw = dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5) if args.config == "c4" else {}
    if args.step_bias is not None:
        cfg_kw = dict(cfg_kw, step_bias=float(args.step_bias))
    cfg = EngineConfig(mfma_dtype=args.mfma, **cfg_kw)
    B = args.batch
    # the engine exactly as AIRonMNIST.train_step builds it (mnist_model.py: per-step canvases kept, as model.py:86-95 exposes them)
    eng = AIREngine(cfg, B, device=device, seed=D.rank_seed(1, rank), keep_canvas_steps=True)
    n_obj = 4 if args.config == "c4" else 2
    imgs, _ = synthetic_multi_mnist(B, cfg.img_size, max_objects=n_obj, seed=rank)
    eng.set_obs(torch.from_numpy(imgs).to(device))
    if not args.fixed_batch:
        # a fresh batch per step: the first node of the captured step gathers it from an HBM-resident synthetic set,
        # each rank from its own Philox stream -- the input pipeline of scripts/multi_mnist.py --device-feeder
        eng.attach_dataset(resident_dataset(cfg.img_size, n_obj, device, n=args.dataset_images, seed=rank), shuffle=True, seed=1,
                           rank=rank, world=world)
    input_note = (("fresh batch per step: air_batch_gather from %d HBM-resident synthetic images per rank, first node of the captured step"
                   % args.dataset_images) if not args.fixed_batch else "one fixed synthetic batch per rank (--fixed-batch)")
    # replicated weights (broadcast from rank 0), one all-reduce (sum) of the flat gradient bucket per step,
    # RMSProp applies grad_scale = 1/world
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    lib = H.lib()
    state = {"dp": None, "done": {}, "ab": {}, "allreduce": None}

    def run_protocol(proto):
        """W warm-up + EXACTLY K timed updates between two barriers (max over ranks) under one data-parallel protocol, then the
        per-update HIP-event median and the replica check.  Returns the record; state['dp'] holds the wrapper."""
        if state["dp"] is not None:
            state["dp"].close()
        dp = D.DataParallelEngine(eng, capture_graph=not args.no_graph, steps_per_replay=args.steps_per_replay, collective=proto)
        state["dp"] = dp
        spr = dp.steps_per_replay                               # 1 unless --steps-per-replay K on a single GPU
        if spr > 1:
            for j in range(1, spr):                             # a different synthetic batch in every slot of the input queue
                eng.set_obs_slot(j, torch.from_numpy(synthetic_multi_mnist(B, cfg.img_size, max_objects=4 if args.config == "c4" else 2,
                                                                          seed=rank + 100 * j)[0]).to(device))
            args.steps = (args.steps + spr - 1) // spr * spr
            args.warmup = (args.warmup + spr - 1) // spr * spr
        if os.environ.get("AIR_BENCH_FAKE_HANG", "") == dp.collective:      # test aid: a protocol that never returns
            time.sleep(10 ** 6)
        with quiet_host():
            for _ in range(args.warmup // spr):
                dp.train_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps // spr):
                dp.train_step()
            barrier()
            elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        finite = bool(torch.isfinite(eng.flat_params).all().item())
        eng.synchronize()
        # the model state the last timed step ran in: the canvas kernels cost what the glimpses cover (the scales of `where`)
        mstate = {"steps_present_per_image": round(float(eng.presence.sum(0).mean().item()), 3),
                  "mean_abs_where": [round(float(v), 3) for v in eng.where.abs().mean(dim=(0, 1)).tolist()]}
        in_sync = dp.replicas_in_sync()      # (collective; outside the timed region) every rank ended with the same bits
        # SURVEY 8(d) asks for the median step time: HIP events around every step of a second, shorter run on the engine stream (the
        # headline `value` stays "exactly K steps between two barriers", as the driver contract defines it).  EVERY rank runs these
        # steps -- with world > 1 each of them contains the gradient all-reduce, which one rank cannot enter alone.
        sp = eng._sp()
        n_ev = min(args.steps // spr, 400)
        evs = [ctypes.c_void_p() for _ in range(n_ev + 1)]
        for e in evs:
            _lib.check(lib.air_event_create(ctypes.byref(e)))
        _lib.check(lib.air_event_record(evs[0], sp))
        for i in range(n_ev):
            dp.train_step()
            _lib.check(lib.air_event_record(evs[i + 1], sp))
        eng.synchronize()
        per = []
        for i in range(n_ev):
            ms = ctypes.c_float()
            _lib.check(lib.air_event_elapsed_ms(evs[i], evs[i + 1], ctypes.byref(ms)))
            per.append(ms.value)
        for e in evs:
            lib.air_event_destroy(e)
        per.sort()
        barrier()
        return {"collective": dp.collective, "elapsed": elapsed, "finite": finite, "in_sync": in_sync, "spr": spr, "model_state": mstate,
                "median_ms": per[len(per) // 2] / spr, "ms_per_step": elapsed / args.steps * 1e3,
                "value": world * B * args.steps / elapsed, "rccl_nranks": dp.rccl_nranks}

    def bare_allreduce():
        """the step's one collective on its own: the flat gradient bucket, summed over the ranks, on the engine stream"""
        nbytes = eng.flat_grads.numel() * 4
        buf = torch.zeros_like(eng.flat_grads)
        with eng.stream_context():
            for _ in range(5):
                dist.all_reduce(buf)
        barrier()
        reps = 30
        t0 = time.perf_counter()
        with eng.stream_context():
            for _ in range(reps):
                dist.all_reduce(buf)
        eng.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        us = t.item() / reps * 1e6
        busbw = 2.0 * (world - 1) / world * nbytes / (us * 1e-6) / 1e9         # nccl-tests' bus bandwidth of an all-reduce
        # per-GPU xGMI injection: 7 links x ~153 GB/s (the prompt's figure; each link is point to point, so a ring is bound by ONE
        # link per hop and only a direct / tree algorithm over all seven can approach this)
        peak = 7 * 153.0
        return {"bytes": nbytes, "us": round(us, 1), "algbw_GBs": round(nbytes / (us * 1e-6) / 1e9, 1), "busbw_GBs": round(busbw, 1),
                "roofline_comm": {"bound": "xgmi", "achieved": round(busbw, 1), "peak": peak, "unit": "GB/s", "frac": round(busbw / peak, 4),
                                  "note": "bus bandwidth 2(N-1)/N x bytes / time of the 10.5 MB gradient all-reduce alone; peak = 7 xGMI links "
                                          "x 153 GB/s per GPU"}}

    # ---- N > 1: the safe protocol first, then the bare collective, then the overlapped protocol under a watchdog --------------------
    # First contact with a multi-GPU node must yield a line whatever happens: `torch-split` (graph | all-reduce | graph, nothing in
    # flight concurrently) is measured and validated FIRST; if anything later hangs -- the overlapped protocol has never met RCCL
    # over xGMI with more than one rank -- every rank's own watchdog thread lets rank 0 print the line of the already finished
    # measurement and ends the process (exit code 0) instead of blocking until the driver's timeout.
    import threading
    limit_s = float(os.environ.get("AIR_BENCH_PROTOCOL_TIMEOUT_S", "0") or 0)
    wd = {"armed": None, "deadline": None, "fallback": None}

    def watchdog():
        while True:
            time.sleep(0.25)
            if wd["armed"] is not None and time.perf_counter() > wd["deadline"]:
                if rank == 0 and wd["fallback"] is not None:
                    wd["fallback"]("%s did not finish within %.0f s: the line is the %s measurement" % (
                        wd["armed"], wd["limit"], state["done"].get("headline", {}).get("collective")))
                sys.stdout.flush()
                os._exit(0 if wd["fallback"] is not None else 3)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    def guarded(name, fn, limit):
        wd["limit"] = limit
        wd["deadline"] = time.perf_counter() + limit
        wd["armed"] = name
        try:
            return fn()
        finally:
            wd["armed"] = None

    fixed_rec = None
    if world == 1:
        head = run_protocol(None)
        protocol_ab = None
        if not args.fixed_batch and not args.no_graph:
            # rounds 1-5's input beside it, for continuity: the same engine on ONE fixed batch (it collapses onto it within ~50 updates)
            eng.attach_dataset(None)
            eng.set_obs(torch.from_numpy(imgs).to(device))
            keep = (args.steps, args.warmup)
            fr = run_protocol(None)
            args.steps, args.warmup = keep
            fixed_rec = {"ms_per_step": round(fr["ms_per_step"], 4), "median_ms_per_step": round(fr["median_ms"], 4), "value": round(fr["value"], 1),
                         "kernel_launches_per_step": sum(eng.kernel_launch_count().values()), "model_state_at_end": fr["model_state"],
                         "input": "one fixed synthetic batch (rounds 1-5's measurement), run after the headline on the same engine"}
            eng.attach_dataset(resident_dataset(cfg.img_size, n_obj, device, n=args.dataset_images, seed=rank), shuffle=True, seed=1)
            state["dp"].close()
            state["dp"] = D.DataParallelEngine(eng, capture_graph=not args.no_graph, steps_per_replay=args.steps_per_replay, collective=None)
    else:
        first_limit = limit_s if limit_s > 0 else 600.0
        head = guarded("torch-split", lambda: run_protocol("torch-split"), first_limit)
        state["done"]["headline"] = head
        state["ab"]["torch-split"] = head
        # from here on a hang costs nothing: the fallback prints torch-split's line
        later_limit = limit_s if limit_s > 0 else max(120.0, 20.0 * (head["elapsed"] * (1 + args.warmup / max(args.steps, 1)) * 2 + 10.0))
        line_holder = {}
        wd["fallback"] = lambda why: print(json.dumps(line_holder["make"](state["done"]["headline"], why)), flush=True)
        line_holder["make"] = None          # set below, once the line builder exists
        state["line_holder"] = line_holder
    elapsed = head["elapsed"]

    (Hh, Ww), (hh, ww) = cfg.img_size, cfg.crop_size
    named = {"c2": "BASELINE configs[1]", "c4": "BASELINE configs[3]", "c5": "BASELINE configs[4]"}[args.config]
    if (args.config, B, args.mfma) not in (("c2", 64, "f32"), ("c4", 64, "f32"), ("c5", 1024, "bf16")):
        named += " shapes at a non-default batch / precision"
    workload = (f"multi-MNIST-shaped {Hh}x{Ww} canvas, max_steps={eng.T}, glimpse {hh}x{ww}, batch={B} per GPU, "
                f"{'fp32' if args.mfma == 'f32' else 'bf16-operand MFMA MLP path'}, "
                f"{'hipGraph replay' if not args.no_graph else 'eager launches'} ({named})")

    def make_line(rec, note=None):
        """the JSON line for the measurement `rec` -- everything that needs no further GPU work (the watchdog prints this form)"""
        ab = {k: {"images_per_sec": round(v["value"], 1), "ms_per_step": round(v["ms_per_step"], 4),
                  "median_ms_per_step": round(v["median_ms"], 4), "replicas_in_sync_after_run": v["in_sync"],
                  "params_finite_after_run": v["finite"]} for k, v in state["ab"].items()} if world > 1 else None
        ar = state["allreduce"]
        line = {
            "metric": "images/sec (train step, ELBO backward) multi-MNIST 50x50, 3-step AIR" if args.config != "c4"
                      else "images/sec (train step, ELBO backward) 100x100 canvas, 5-step AIR, glimpse 28x28",
            "value": round(rec["value"], 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(rec["ms_per_step"], 4), "median_ms_per_step": round(rec["median_ms"], 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.mfma == "f32" else "bf16 operands / f32 accumulate+storage",
            "data": "synthetic" if not share_gpu else "synthetic; NOT A MEASUREMENT: all ranks share one GPU (AIR_BENCH_SHARE_GPU)",
            "input": input_note,
            "config": {"workload": workload if args.step_bias is None else workload + " -- PROBE: --step-bias %g" % args.step_bias,
                       "step_bias": cfg.step_bias, "global_batch": world * B, "batch_per_gpu": B, "parallelism": f"dp{world}",
                       "hipgraph": not args.no_graph, "steps_per_graph_replay": rec["spr"],
                       "kernel_launches_per_step": sum(eng.kernel_launch_count().values()),
                       "kernel_launches_by_lane": eng.kernel_launch_count(),
                       "keep_canvas_steps": True, "collective": rec["collective"],
                       # N > 1: every protocol timed in this invocation (the headline is the fastest one whose replicas ended bit-identical
                       # and finite), and the step's one collective on its own
                       "protocol_ab": ab, "protocol_note": note,
                       # the prediction this measurement is there to falsify, with what this run measured beside it:
                       # efficiency = single-GPU images/s per GPU (the default line of the same binary) / this run's per-GPU rate
                       "protocol_prediction_at_8_ranks": PREDICTED_AT_8_RANKS if world > 1 else None,
                       "protocols_opt_in": None if world == 1 else "ipc-rsag (AIR_BENCH_PROTOCOLS=torch-overlap,ipc-rsag): validated with two processes on one GPU only",
                       "allreduce_us": ar["us"] if ar else None, "allreduce_bytes": ar["bytes"] if ar else None,
                       "allreduce_busbw_GBs": ar["busbw_GBs"] if ar else None,
                       # ranks as the communication layer itself reports them: ncclCommCount of the engine's own communicator
                       # (rccl-split / rccl-captured), and the size of torch.distributed's process group (backend nccl = RCCL)
                       "rccl_nranks": rec["rccl_nranks"] if rec["rccl_nranks"] is not None else (dist.get_world_size() if world > 1 and not share_gpu else None),
                       "dist_world_size": (dist.get_world_size() if world > 1 else 1),
                       "dist_backend": (dist.get_backend() if world > 1 else None),
                       "params_finite_after_run": rec["finite"], "replicas_in_sync_after_run": rec["in_sync"],
                       "model_state_at_end": rec.get("model_state"),
                       # HIP runtime settings the package put into the environment before the runtime initialised
                       # (attend_infer_repeat_amd/runtime_env.py; a user's own export wins; "late": torch had initialised HIP first)
                       "hip_runtime_env": dict(_runtime_env.applied, late=_runtime_env.late)},
            "roofline": None, "roofline_comm": ar["roofline_comm"] if ar else None, "cpu_baseline": None,
        }
        return line

    if world > 1:
        state["line_holder"]["make"] = make_line
        ar = guarded("the bare gradient all-reduce", bare_allreduce, later_limit)
        state["allreduce"] = ar
        # Protocols timed after the safe one.  "ipc-rsag" (round 5: barrier | shard sum + sharded RMSProp + parameter push | barrier as
        # kernel nodes over hipIpc-mapped peer buffers, no library collective) has only ever run with two processes on ONE GPU; a
        # wrong peer mapping on real multi-GPU hardware would be a GPU fault, which no watchdog survives -- so on real GPUs it is
        # opt-in (AIR_BENCH_PROTOCOLS=torch-overlap,ipc-rsag) and the default run cannot lose its line to it.
        default_protocols = "torch-overlap,ipc-rsag" if share_gpu else "torch-overlap"
        extra = [x for x in os.environ.get("AIR_BENCH_PROTOCOLS", default_protocols).split(",") if x and x != "torch-split"]
        notes = []
        for proto in extra:
            rec = guarded(proto, lambda: run_protocol(proto), later_limit)
            if rec["collective"] in state["ab"]:
                notes.append("%s is not available for this plan (it runs as %s)" % (proto, rec["collective"]))
                continue
            state["ab"][rec["collective"]] = rec
            if rec["finite"] and rec["in_sync"] and rec["value"] > head["value"]:
                head = rec
                state["done"]["headline"] = head
            elif not (rec["finite"] and rec["in_sync"]):
                notes.append("%s is NOT VALIDATED (replicas in sync: %s, finite: %s): not eligible for the headline" % (
                    rec["collective"], rec["in_sync"], rec["finite"]))
        protocol_note = "; ".join(notes) or None
    else:
        protocol_note = None

    if rank == 0:
        def full_line():
            line = make_line(head, protocol_note)
            if fixed_rec is not None:
                line["fixed_batch"] = fixed_rec
            if args.breakdown:
                plan_breakdown(eng)
            roof = st_rooflines(eng)
            # the dominant ST kernel AS IT RUNS IN THE TIMED STEP: attend_fwd_kernel (the fused affine-grid + bilinear glimpse read of
            # all T steps + the tiny heads around it), launched from the step's own plan entry on the step's buffers and timed with
            # HIP events on the engine stream; charged with the read's SURVEY 8(d) bytes only.  The stand-alone read kernel
            # (st_read_fwd_lean_kernel, what the sweeps scale out of cache) is next to it in roofline_standalone_read.
            if "attend_fwd" in roof and not args.no_graph:
                # primary figure = the IN-STEP duration: the committed rocprofv3 positions of this binary,
                # shape and plan when there is one, else HIP events around the entry inside the eagerly issued step; the warm
                # back-to-back launch time (what earlier rounds printed as `frac`) is kept as `frac_live`
                line["roofline"] = attend_roofline(eng, lib, live_warm_us=roof["attend_fwd"]["us_per_launch"])
                line["roofline"]["traffic"] = roof["attend_fwd"].get("traffic")
                line["roofline"]["minimal_bytes_per_launch"] = roof["attend_fwd"]["minimal_bytes_per_launch"]
                line["roofline"]["definition_note"] = ("since round 4 the headline roofline is the in-step fused kernel (attend_fwd_kernel) charged with "
                                                       "the read's 8(d) bytes only; rounds 1-3 printed the stand-alone read kernel: fractions are not "
                                                       "comparable across that change; since round 5 `frac` is the in-step duration, `frac_live` the warm one")
            elif "attend_fwd" in roof:
                line["roofline"] = dict(roof["attend_fwd"], kernel="attend_fwd_kernel, eager launches (--no-graph): back-to-back HIP-event time")
            else:
                line["roofline"] = dict(roof["st_read_fwd"], kernel="st_read_fwd_lean_kernel (this plan has no fused attend launch: the read runs on its own)")
            if plan_env_overrides():
                line["config"]["plan_env_overrides"] = plan_env_overrides()
            line["roofline_standalone_read"] = dict(roof["st_read_fwd"], kernel="st_read_fwd_lean_kernel launched on its own at the in-step shape")
            line["roofline_other_kernels"] = {k: v for k, v in roof.items() if k not in ("st_read_fwd", "attend_fwd")}
            line["roofline_gemm"] = gemm_roofline(eng)
            if not args.no_sweep and world == 1:
                # T glimpses per staged image (as in the train step) and the 1:1 case (one image per glimpse); `frac` is computed
                # from the bytes the launch must move, so it cannot exceed 1
                line["roofline_sweep_st_read_fwd"] = st_read_sweep(cfg, eng.T, [64, 512, 1024, 8192, 65536], device)
                line["roofline_sweep_st_read_fwd_one_image_per_glimpse"] = st_read_sweep(cfg, 1, [192, 3072, 24576, 196608],
                                                                                         device, share_image=False)
                cw_f, cw_b, cw_i = canvas_write_sweep(cfg, eng.T, [64, 1024, 8192, 65536], device)
                line["roofline_sweep_canvas_write_fwd"], line["roofline_sweep_canvas_write_bwd"] = cw_f, cw_b
                line["roofline_sweep_canvas_write_pair"] = cw_i
                line["stream_reference"] = stream_reference(device)
            if (world == 1 and not args.no_other_configs and not args.no_graph and (args.config, B, args.mfma) == ("c2", 64, "f32")):
                # the other named single-GPU configurations, driver-observed: ~2 s each
                line["other_configs"] = {}
                for oc in ("c4", "c5"):
                    try:
                        line["other_configs"][oc] = run_other_config_seeds(oc, device)
                    except Exception as ex:             # noqa: BLE001 -- the headline line must survive whatever happens here
                        line["other_configs"][oc] = {"error": repr(ex)}
            if not args.no_cpu_baseline and world == 1:
                torch.cuda.synchronize(device)
                line["cpu_baseline"] = cpu_baseline(cfg_kw, B, args.cpu_seconds, args.cpu_all_cores)
            return line
        line = guarded("the per-kernel rooflines", full_line, later_limit) if world > 1 else full_line()
        wd["fallback"] = None
        print(json.dumps(line), flush=True)
    wd["fallback"] = None
    if world > 1:
        # the other ranks wait here for rank 0's stand-alone kernel timings; unbounded patience is fine: rank 0 is guarded
        wd["armed"] = None
        dist.barrier()
    state["dp"].close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
