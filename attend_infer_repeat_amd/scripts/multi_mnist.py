#!/usr/bin/env python
"""Counterpart of the reference's training script (attend_infer_repeat/scripts/multi_mnist.py:24-147): same
hyper-parameters, same loop structure (train / periodic log / periodic figure + checkpoint), on the MI355X engine.

Data: `--data-dir` with mnist_train.pickle / mnist_validation.pickle (Python-3 pickles with the reference's layout) if
present; otherwise a synthetic multi-MNIST-shaped dataset (no network here, so no MNIST download).  The dataset lives
in HBM and batches are index gathers (data.DeviceFeeder) instead of a tf.py_func host round trip per step.
"""
import argparse
import json
import os
import os.path as osp
import sys
import time

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from attend_infer_repeat_amd.data import DeviceFeeder, load_data, procedural_multi_mnist, synthetic_dataset  # noqa: E402
from attend_infer_repeat_amd.evaluation import make_fig, make_logger, step_summaries  # noqa: E402
from attend_infer_repeat_amd.mnist_model import AIRonMNIST  # noqa: E402
from attend_infer_repeat_amd.utils import AttrDict  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=int(3e5))            # multi_mnist.py:134
    ap.add_argument("--log-every", type=int, default=10000)           # :142
    ap.add_argument("--save-every", type=int, default=5000)           # :145
    ap.add_argument("--results-dir", default="../results")
    ap.add_argument("--run-name", default="multi_mnist")
    ap.add_argument("--data-dir", default="data")
    ap.add_argument("--synthetic-samples", type=int, default=60000)
    ap.add_argument("--eval-batches", type=int, default=10)
    ap.add_argument("--figures", action="store_true")
    ap.add_argument("--summary-every", type=int, default=1000)        # multi_mnist.py:138-140
    ap.add_argument("--glyphs", action="store_true",
                    help="no multi-MNIST pickles: synthesise the dataset with the reference's generator (data.create_multi_mnist) "
                         "from procedural digit templates instead of stroke blobs")
    ap.add_argument("--learning-rate", type=float, default=1e-4)
    ap.add_argument("--check-every", type=int, default=0,
                    help="diagnostics (read-only: does not touch the noise stream): every N updates from --check-from on, write the "
                         "extreme values of the latents' scales, of the parameters, gradients and RMSProp slots to log.jsonl")
    ap.add_argument("--check-from", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device-feeder", action="store_true",
                    help="draw every training batch inside the captured step from the HBM-resident training set (engine "
                         "attach_dataset): no host work between updates")
    ap.add_argument("--guard-degenerate", type=float, default=0.0,
                    help="stability switch, default off (= the reference's arithmetic): floor of both Gaussian heads' scale and "
                         "minimum |scale component| of the sampled `where` (AIRonMNIST(guard_degenerate=...); e.g. 1e-6)")
    ap.add_argument("--resume", default=None,
                    help="checkpoint written by this script (model_<iter>.pt): restores parameters, the RMSProp slots, "
                         "the step counter, the learning rate, the Philox noise state and the feeders' positions")
    ap.add_argument("--init-from-tf-ckpt", default=None, metavar="PREFIX",
                    help="prefix of a checkpoint the reference's tf.train.Saver wrote (e.g. ../results/multi_mnist/model.ckpt-175000): "
                         "the model / baseline variables become the initial parameters, its RMSProp slots (where the file holds them) the "
                         "optimiser state and its global_step the step counter (tf_checkpoint.py: format and names unvalidated without TensorFlow)")
    ap.add_argument("--grad-histograms", action="store_true",
                    help="add the per-variable gradient histograms of evaluation.gradient_summaries(histogram=True) (the reference's default, "
                         "evaluation.py:221-248) to the 1000-iteration summaries in log.jsonl")
    ap.add_argument("--tf-name-map", default=None, metavar="JSON",
                    help="with --init-from-tf-ckpt: a JSON file {engine parameter name: checkpoint variable name} that replaces the shape-based "
                         "matcher (tf_checkpoint.default_name_map) when it stops or guesses wrong")
    args = ap.parse_args(argv)

    learning_rate, n_steps, batch_size = args.learning_rate, 3, 64    # multi_mnist.py:24-25,37
    num_steps_prior = AttrDict(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3)
    appearance_prior = AttrDict(loc=0., scale=1.)
    where_scale_prior = AttrDict(loc=0., scale=1.)
    where_shift_prior = AttrDict(loc=0., scale=1.)
    step_bias, transform_var_bias, output_multiplier, init_explore_eps, l2_weight = .75, .5, .5, 1e-3, 0.

    device = torch.device("cuda", 0)
    torch.manual_seed(args.seed)                                      # parameter initialisation (Sonnet draws at build time)
    logdir = osp.join(args.results_dir, args.run_name)
    os.makedirs(logdir, exist_ok=True)
    tr, va = osp.join(args.data_dir, "mnist_train.pickle"), osp.join(args.data_dir, "mnist_validation.pickle")
    if osp.exists(tr) and osp.exists(va):
        train_data, valid_data = load_data(tr), load_data(va)
    elif args.glyphs:
        print("no multi-MNIST pickles under {!r}: procedural digit templates through the reference's generator".format(args.data_dir))
        as_float = lambda d: dict(imgs=d["imgs"].astype("float32") / 255.0, nums=d["nums"].astype("float32"))
        train_data = as_float(procedural_multi_mnist(args.synthetic_samples, seed=args.seed))
        valid_data = as_float(procedural_multi_mnist(max(args.synthetic_samples // 6, batch_size), seed=args.seed + 1000))
    else:
        print("no multi-MNIST pickles under {!r}: using a synthetic dataset".format(args.data_dir))
        train_data = synthetic_dataset(args.synthetic_samples, seed=args.seed)
        valid_data = synthetic_dataset(max(args.synthetic_samples // 6, batch_size), seed=args.seed + 1)
    train_feed = DeviceFeeder(train_data, batch_size, device, shuffle=True, seed=args.seed)
    valid_feed = DeviceFeeder(valid_data, batch_size, device, shuffle=False)
    x, y = train_feed()

    n_hiddens = [32 * 8] * 2
    air = AIRonMNIST(x, y, max_steps=n_steps, explore_eps=init_explore_eps, inpt_encoder_hidden=n_hiddens,
                     glimpse_encoder_hidden=n_hiddens, glimpse_decoder_hidden=n_hiddens,
                     transform_estimator_hidden=n_hiddens, steps_pred_hidden=[128, 64], baseline_hidden=[256, 128],
                     transform_var_bias=transform_var_bias, step_bias=step_bias, output_multiplier=output_multiplier,
                     guard_degenerate=args.guard_degenerate or None)
    train_step, global_step = air.train_step(learning_rate, l2_weight, appearance_prior, where_scale_prior,
                                             where_shift_prior, num_steps_prior)
    if args.init_from_tf_ckpt:
        from attend_infer_repeat_amd.tf_checkpoint import global_step_of, import_tf_checkpoint, import_tf_optimizer_slots, mapping_report
        name_map = None
        if args.tf_name_map:
            name_map = json.load(open(args.tf_name_map))
        named = import_tf_checkpoint(args.init_from_tf_ckpt, air._engine.param_shapes, name_map=name_map)
        air._engine.load_parameters({k: torch.from_numpy(v) for k, v in named.items()})
        air._engine.reset_optimizer()
        slots = import_tf_optimizer_slots(args.init_from_tf_ckpt, air._engine.param_shapes, name_map=name_map)
        left = mapping_report(args.init_from_tf_ckpt, air._engine.param_shapes, name_map)
        if left["unmapped_engine_parameters"]:
            print('WARNING: not in the checkpoint (kept at their initial values): {}'.format(', '.join(left["unmapped_engine_parameters"])))
        if left["unused_checkpoint_variables"]:
            print('WARNING: checkpoint variables nobody used (pass --tf-name-map if one of them belongs to the model): {}'.format(
                ', '.join(left["unused_checkpoint_variables"])))
        half = [n for n in named if n not in slots['ms']]
        if half and slots['ms']:
            print('WARNING: no complete RMSProp slots in the checkpoint for: {}'.format(', '.join(sorted(half))))
        air._engine.load_optimizer_slots(**{k: {n: torch.from_numpy(v) for n, v in d.items()} for k, d in slots.items()})
        step0 = global_step_of(args.init_from_tf_ckpt)
        if step0 is not None:
            air._engine.set_global_step(step0); air.global_step.fill_(step0)
        print('Initialised {} tensors (+ RMSProp slots of {}) from {} (global_step {})'.format(len(named), len(slots['ms']), args.init_from_tf_ckpt, step0))
        global_step = air.global_step
    if args.resume:
        # the reference only ever saves (tf.train.Saver over every variable incl. the optimiser slots and global_step,
        # multi_mnist.py:116,145-146); restoring is the counterpart a long run needs
        ck = torch.load(args.resume, map_location="cpu")
        air._engine.load_state_dict(ck["engine"])
        air.global_step.fill_(int(ck["engine"]["global_step"]))
        if "train_feed" in ck:
            train_feed.load_state_dict(ck["train_feed"]); valid_feed.load_state_dict(ck["valid_feed"])
        global_step = air.global_step
    if args.device_feeder:
        air._engine.attach_dataset(train_feed.imgs.reshape(train_feed.n, -1), shuffle=True, seed=args.seed)
        air._engine.capture()
    writer = open(osp.join(logdir, "log.jsonl"), "a")
    log = make_logger(air, train_feed, args.eval_batches, valid_feed, args.eval_batches, writer)

    train_itr = int(global_step)
    print('Starting training at iter = {}'.format(train_itr))
    if train_itr == 0:
        log(0)
    t0, last = time.time(), train_itr
    while train_itr < args.iters:
        if args.device_feeder:
            train_itr = int(train_step(None, None, refresh=False))
        else:
            xb, yb = train_feed()
            train_itr = int(train_step(xb, yb, refresh=False))
        if args.check_every and train_itr >= args.check_from and train_itr % args.check_every == 0:
            eng = air._engine
            o = eng.outputs()
            fin = lambda t: bool(torch.isfinite(t).all().item())
            diag = dict(step=train_itr, data="check",
                        where_scale_min=float(o["where_scale"].min()), where_scale_max=float(o["where_scale"].max()),
                        what_scale_min=float(o["what_scale"].min()), what_scale_max=float(o["what_scale"].max()),
                        where_abs_max=float(o["where"].abs().max()), what_abs_max=float(o["what"].abs().max()),
                        presence_prob_min=float(o["presence_prob"].min()), presence_prob_max=float(o["presence_prob"].max()),
                        num_step=float(o["num_step_per_sample"].mean()), rec=float(o["rec_loss"]),
                        kl_what=float(o["kl_what"]), kl_where=float(o["kl_where"]),
                        imp_var=float(o["imp_weight_var"]), baseline_abs_max=float(o["baseline"].abs().max()),
                        params_abs_max=float(eng.flat_params.abs().max()), grads_abs_max=float(eng.flat_grads.abs().max()),
                        ms_max=float(eng.flat_ms.max()), ms_min=float(eng.flat_ms.min()), mom_abs_max=float(eng.flat_mom.abs().max()),
                        finite=dict(params=fin(eng.flat_params), grads=fin(eng.flat_grads), ms=fin(eng.flat_ms), mom=fin(eng.flat_mom)))
            # the variance slot of centred RMSProp, ms - mg^2, must stay >= 0 for the square root (model.py:265: centered=True)
            diag["centred_var_min"] = float((eng.flat_ms - eng.flat_mg * eng.flat_mg).min())
            worst = {}
            for k, g in eng.named_grads().items():
                worst[k] = float(g.abs().max())
            top = sorted(worst.items(), key=lambda kv: -kv[1] if kv[1] == kv[1] else -float("inf"))[:3]
            diag["largest_grads"] = top
            writer.write(json.dumps(diag) + "\n"); writer.flush()
        if args.summary_every and train_itr % args.summary_every == 0:
            # the reference's `all_summaries` (model.py's tf.summary scalars + evaluation.gradient_summaries), every 1000 iterations
            writer.write(json.dumps(dict(step=train_itr, data="summary", **step_summaries(air, histogram=args.grad_histograms))) + "\n")
        if train_itr % args.log_every == 0:
            torch.cuda.synchronize()
            dt = time.time() - t0
            print("iter {}: {:.0f} images/s".format(train_itr, (train_itr - last) * batch_size / max(dt, 1e-9)))
            log(train_itr)
            t0, last = time.time(), train_itr
        if train_itr % args.save_every == 0:
            torch.save({"engine": air._engine.state_dict(), "train_feed": train_feed.state_dict(),
                        "valid_feed": valid_feed.state_dict()}, osp.join(logdir, "model_{}.pt".format(train_itr)))
            if args.figures:
                make_fig(air, logdir, train_itr)
    writer.close()
    return air


if __name__ == "__main__":
    main()
