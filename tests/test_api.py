"""GPU tests of the drop-in Python surface (AIRCell / AIRModel / AIRonMNIST) -- they read like the reference's own
test/cell_test.py, plus parity against the CPU oracle with injected noise."""
from functools import partial

import numpy as np
import pytest
import torch

from oracle import air_oracle as O

pytestmark = pytest.mark.gpu


def make_modules(amd):                                        # test/cell_test.py:9-18
    return dict(
        transition=amd.rnn.GRU(3),
        input_encoder=(lambda: amd.modules.Encoder(5)),
        glimpse_encoder=(lambda: amd.modules.Encoder(7)),
        glimpse_decoder=(lambda x: amd.modules.Decoder(11, x)),
        transform_estimator=(lambda x: amd.modules.StochasticTransformParam(13, x)),
        steps_predictor=(lambda: amd.modules.StepsPredictor(17)))


@pytest.fixture(scope="module")
def amd(gpu_device):
    import attend_infer_repeat_amd as pkg
    from attend_infer_repeat_amd import cell, mnist_model, model, modules, neural, rnn, utils  # noqa: F401
    return pkg


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_cell_instantiate_like_reference(amd):               # test/cell_test.py:21-63
    batch_size, img_size, crop_size, n_latent, n_steps = 10, (3, 3), (2, 2), 10, 3
    x = torch.rand((batch_size,) + img_size).cuda()
    air = amd.cell.AIRCell(img_size, crop_size, n_latent, **make_modules(amd))
    state = air.initial_state(x)
    outs = []
    for _ in range(n_steps):                                  # tf.nn.dynamic_rnn(air, dummy_sequence, time_major=True)
        o, state = air(None, state)
        outs.append(o)
    outputs = [torch.stack([o[i] for o in outs]) for i in range(10)]
    assert air.output_names == 'canvas glimpse what what_loc what_scale where where_loc where_scale presence_prob presence'.split()
    widths = [9, 4, 10, 10, 10, 4, 4, 4, 1, 1]
    assert [tuple(o.shape) for o in outputs] == [(n_steps, batch_size, wd) for wd in widths]
    assert [int(s) if not isinstance(s, tuple) else s for s in air.output_size] == widths
    canvas = outputs[0].reshape((n_steps, batch_size) + img_size)
    loss = 0.5 * ((x - canvas[-1]) ** 2).sum()                # tf.nn.l2_loss
    opt = torch.optim.Adam(air.parameters(), 1e-4)
    loss.backward(); opt.step()
    assert np.isfinite(loss.item())
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in air.parameters())


def _load_oracle_params(amd, cell, baseline, params, transition):
    """copy oracle-named tensors into the lazily-built modules"""
    def mlp(prefix, m):
        for i, layer in enumerate(m.layers):
            layer.w.data.copy_(params[f"{prefix}/{i}/w"]); layer.b.data.copy_(params[f"{prefix}/{i}/b"])
    mlp("input_encoder", cell._input_encoder.mlp); mlp("transform", cell._transform_estimator.mlp)
    mlp("steps", cell._steps_predictor.mlp); mlp("glimpse_encoder", cell._glimpse_encoder.mlp)
    mlp("glimpse_decoder", cell._glimpse_decoder.mlp)
    cell._what_distrib.w.data.copy_(params["what/w"]); cell._what_distrib.b.data.copy_(params["what/b"])
    tr = cell._transition
    if transition == "lstm":
        tr.w_gates.data.copy_(params["lstm/w_gates"]); tr.b_gates.data.copy_(params["lstm/b_gates"])
        tr.h0.data.copy_(params["lstm/h0"]); tr.c0.data.copy_(params["lstm/c0"])
    else:
        for g in "zrh":
            for k in "wub":
                getattr(tr, f"{k}{g}").data.copy_(params[f"gru/{k}{g}"])
        tr.h0.data.copy_(params["gru/h0"])
    if baseline is not None:
        mlp("baseline", baseline.mlp)


def _build_model(amd, ocfg, obs, transition):
    M = amd.modules
    tr = amd.rnn.LSTM(ocfg.n_hidden) if transition == "lstm" else amd.rnn.GRU(ocfg.n_hidden)
    model = amd.model.AIRModel(
        obs, None, ocfg.max_steps, ocfg.crop_size, ocfg.n_appearance, tr,
        input_encoder=partial(M.Encoder, list(ocfg.inpt_encoder_hidden)),
        glimpse_encoder=partial(M.Encoder, list(ocfg.glimpse_encoder_hidden)),
        glimpse_decoder=partial(M.Decoder, list(ocfg.glimpse_decoder_hidden)),
        transform_estimator=partial(M.StochasticTransformParam, list(ocfg.transform_estimator_hidden),
                                    scale_bias=ocfg.transform_var_bias),
        steps_predictor=partial(M.StepsPredictor, list(ocfg.steps_pred_hidden), ocfg.step_bias),
        output_std=ocfg.output_std, output_multiplier=ocfg.output_multiplier, explore_eps=ocfg.explore_eps)
    return model


@pytest.mark.parametrize("transition", ["lstm", "gru"])
def test_model_unroll_matches_oracle(amd, transition):
    ocfg = O.tiny_config(transition=transition, step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5,
                         output_std=0.3, transform_var_bias=0.5, n_hidden=6)
    B = 10
    params = O.init_params(ocfg, seed=4, bias_std=0.2)
    obs, _ = O.synthetic_batch(ocfg, B, seed=5); obs = torch.rand(B, *ocfg.img_size)
    noise = O.make_noise(ocfg, B, seed=6)
    model = _build_model(amd, ocfg, obs.cuda(), transition)
    _load_oracle_params(amd, model.cell, None, params, transition)
    model.forward(noise={k: v.cuda() for k, v in noise.items()})
    ref = O.unroll({k: v.double() for k, v in params.items()}, ocfg, obs.double(), {k: v.double() for k, v in noise.items()})
    assert torch.equal(model.presence.cpu().double(), ref["presence"])
    for k in ["what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob", "canvas",
              "final_canvas", "glimpse", "num_step_per_sample"]:
        assert rel(getattr(model, k).reshape(ref[k].shape), ref[k]) < 1e-4, k
    q = model.num_steps_distrib.prob()
    assert rel(q, O.bernoulli_to_modified_geometric(ref["presence_prob"].reshape(ocfg.max_steps, B).t())) < 1e-5


def test_generic_train_step_matches_oracle(amd):
    """AIRModel (autograd over the HIP kernels) with an LSTM: losses, both gradient sets and one RMSProp update."""
    ocfg = O.AIRConfig(img_size=(12, 10), crop_size=(5, 4), n_appearance=6, n_hidden=16, inpt_encoder_hidden=(24,),
                       glimpse_encoder_hidden=(20,), glimpse_decoder_hidden=(18,), transform_estimator_hidden=(14,),
                       steps_pred_hidden=(9,), baseline_hidden=(12, 7), max_steps=3)
    B = 9
    params = O.init_params(ocfg, seed=7, bias_std=0.2)
    obs = torch.rand(B, *ocfg.img_size)
    noise = O.make_noise(ocfg, B, seed=8)
    AD = amd.utils.AttrDict
    model = _build_model(amd, ocfg, obs.cuda(), "lstm")
    baseline = amd.modules.BaselineMLP(list(ocfg.baseline_hidden))
    nsp = AD(anneal='exp', init=ocfg.nsp_init, final=ocfg.nsp_final, steps_div=ocfg.nsp_steps_div,
             steps=ocfg.nsp_steps, hold_init=ocfg.nsp_hold_init)
    train_step, global_step = model.train_step(ocfg.learning_rate, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                                               AD(loc=0., scale=1.), nsp, baseline=baseline)
    _load_oracle_params(amd, model.cell, model.baseline_module, params, "lstm")
    gnoise = {k: v.cuda() for k, v in noise.items()}
    train_step(noise=gnoise)
    assert int(global_step) == 1
    p64 = {k: v.double() for k, v in params.items()}
    res, grads = O.forward_backward(p64, ocfg, obs.double(), {k: v.double() for k, v in noise.items()}, global_step=0)
    for k in ["rec_loss", "kl_num_steps", "kl_what", "kl_where", "reinforce_loss", "baseline_loss", "opt_loss"]:
        assert abs(getattr(model, k).item() - res[k].item()) < 2e-4 * (abs(res[k].item()) + 1), k
    assert rel(model.loss.value, res["loss"]) < 2e-4 and rel(model.loss.per_sample, res["loss_per_sample"]) < 2e-4
    assert tuple(model.importance_weight.shape) == (B, B)                     # the reference's broadcast quirk
    named = {"what/w": model.cell._what_distrib.w, "lstm/w_gates": model.cell._transition.w_gates,
             "lstm/h0": model.cell._transition.h0,
             "input_encoder/0/w": model.cell._input_encoder.mlp.layers[0].w,
             "transform/1/b": model.cell._transform_estimator.mlp.layers[1].b,
             "steps/1/w": model.cell._steps_predictor.mlp.layers[1].w,
             "glimpse_decoder/1/w": model.cell._glimpse_decoder.mlp.layers[1].w,
             "baseline/0/w": model.baseline_module.mlp.layers[0].w, "baseline/2/b": model.baseline_module.mlp.layers[2].b}
    for k, p in named.items():
        assert rel(p.grad, grads[k]) < 2e-3, (k, rel(p.grad, grads[k]))
    slots = O.rmsprop_init(p64)
    O.rmsprop_centered_step(p64, grads, slots, ocfg)
    for k, p in named.items():
        assert rel(p.data.cpu().double() - params[k].double(), p64[k] - params[k].double()) < 5e-3, k


def test_aironmnist_engine_backed_training(amd):
    """The reference script's model (scripts/multi_mnist.py:82-100) with the fused engine behind train_step."""
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    AD = amd.utils.AttrDict
    B = 16
    imgs, nums = synthetic_multi_mnist(B, (50, 50), 2, seed=0)
    x, y = torch.from_numpy(imgs).cuda(), torch.from_numpy(nums).cuda()
    n_hiddens = [256, 256]
    air = amd.mnist_model.AIRonMNIST(x, y, max_steps=3, explore_eps=1e-3, inpt_encoder_hidden=n_hiddens,
                                     glimpse_encoder_hidden=n_hiddens, glimpse_decoder_hidden=n_hiddens,
                                     transform_estimator_hidden=n_hiddens, steps_pred_hidden=[128, 64],
                                     baseline_hidden=[256, 128], transform_var_bias=.5, step_bias=.75,
                                     output_multiplier=.5)
    nsp = AD(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3)
    train_step, global_step = air.train_step(1e-4, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                                             AD(loc=0., scale=1.), nsp)
    assert air._engine is not None
    w_before = air.cell._input_encoder.mlp.layers[0].w.detach().clone()
    for _ in range(3):
        train_step()
    assert int(global_step) == 3 and air._engine.step_dev.item() == 3
    assert air.canvas.shape == (3, B, 50, 50) and air.glimpse.shape == (3, B, 20, 20)
    assert air.what.shape == (3, B, 50) and air.where.shape == (3, B, 4) and air.presence.shape == (3, B, 1)
    assert air.rec_loss_per_sample.shape == (B,) and np.isfinite(air.opt_loss.item())
    assert 0.0 <= air.num_step_accuracy.item() <= 1.0
    # module tree and engine share storage: the cell sees the trained weights
    w_after = air.cell._input_encoder.mlp.layers[0].w
    assert w_after.data_ptr() == air._engine.params["input_encoder/0/w"].data_ptr()
    assert not torch.equal(w_before, w_after)
    # stepping the cell by hand with the engine's noise reproduces the engine's forward
    eng = air._engine
    eng.forward(sample_noise=False)
    eo = eng.outputs()
    air.forward(noise=dict(eps_where=eng.eps_where.clone(), eps_what=eng.eps_what.clone(),
                           u_pres=eng.u_pres.clone().unsqueeze(-1)))
    assert torch.equal(air.presence.reshape(3, B), eo["presence"].reshape(3, B))
    assert rel(air.where, eo["where"]) < 1e-4 and rel(air.final_canvas, eo["final_canvas"]) < 1e-4


def test_runtime_switches_reach_the_captured_engine(amd):
    """model.py:58,71,307-308 / mnist_model.py:24-26: use_prior (toggle_prior), explore_eps, step_bias, transform_var_bias and
    output_multiplier are non-trainable VARIABLES of the reference, assignable between steps.  On the engine path they are
    launch arguments of a captured graph; after toggling / assigning them mid-run the next fused step must be the step the
    oracle takes with the new values (same weights, the engine's own noise), not the step of the stale graph."""
    import dataclasses
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    AD = amd.utils.AttrDict
    B = 8
    imgs, nums = synthetic_multi_mnist(B, (50, 50), 2, seed=0)
    x, y = torch.from_numpy(imgs).cuda(), torch.from_numpy(nums).cuda()
    n_hiddens = [256, 256]
    air = amd.mnist_model.AIRonMNIST(x, y, max_steps=3, explore_eps=1e-3, inpt_encoder_hidden=n_hiddens,
                                     glimpse_encoder_hidden=n_hiddens, glimpse_decoder_hidden=n_hiddens,
                                     transform_estimator_hidden=n_hiddens, steps_pred_hidden=[128, 64],
                                     baseline_hidden=[256, 128], transform_var_bias=.5, step_bias=.75,
                                     output_multiplier=.5)
    nsp = AD(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3)
    train_step, _ = air.train_step(1e-4, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.), AD(loc=0., scale=1.), nsp)
    eng = air._engine
    assert eng is not None and eng._graph is not None
    train_step()
    graph_before = eng._graph.value

    def oracle_step_matches(ocfg):
        """the update the engine just made == O.train_step from the parameters before it, with the noise the graph drew"""
        p_before = {k: v.detach().cpu().double().clone() for k, v in eng.params.items()}
        slots = {k: dict(ms=eng.flat_ms[eng.param_offsets[k]:eng.param_offsets[k] + eng.param_sizes[k]].view(eng.param_shapes[k]).cpu().double().clone(),
                         mg=eng.flat_mg[eng.param_offsets[k]:eng.param_offsets[k] + eng.param_sizes[k]].view(eng.param_shapes[k]).cpu().double().clone(),
                         mom=eng.flat_mom[eng.param_offsets[k]:eng.param_offsets[k] + eng.param_sizes[k]].view(eng.param_shapes[k]).cpu().double().clone())
                 for k in eng.params}
        gs = int(eng.step_dev.item())
        train_step()
        eng.synchronize()
        noise = dict(eps_where=eng.eps_where.cpu().double(), eps_what=eng.eps_what.cpu().double(),
                     u_pres=eng.u_pres.cpu().double().reshape(3, B, 1))
        ref = {k: v.clone() for k, v in p_before.items()}
        O.train_step(ref, slots, ocfg, x.cpu().double(), noise, global_step=gs)
        worst = max(rel(eng.params[k].cpu().double() - p_before[k], ref[k] - p_before[k]) for k in ref)
        return worst

    base = O.AIRConfig()
    assert oracle_step_matches(base) < 5e-4                      # the unmodified step agrees (sanity of the harness)
    # toggle the prior off (model.py:308) and move every other switch
    assert air.toggle_prior() is False
    air.explore_eps = 0.05
    air.step_bias.fill_(0.25)
    air.transform_var_bias.fill_(-0.5)
    air.output_multiplier.fill_(0.75)
    changed = dataclasses.replace(base, use_prior=False, explore_eps=0.05, step_bias=0.25, transform_var_bias=-0.5,
                                  output_multiplier=0.75)
    assert oracle_step_matches(changed) < 5e-4
    assert eng.cfg.use_prior is False and eng._graph is not None and eng._graph.value != graph_before     # re-captured
    assert oracle_step_matches(base) > 5e-3                      # ... and the stale configuration would NOT have matched
    # the generic cell-by-cell path sees the same switches (shared variables)
    assert air.cell._explore_eps == 0.05
    air.toggle_prior()
    assert oracle_step_matches(dataclasses.replace(changed, use_prior=True)) < 5e-4


def test_aironmnist_bf16_mfma_option(amd):
    """BASELINE configs[4] through the reference's surface: train_step(..., mfma_dtype="bf16") runs the engine with
    bf16-rounded operands; a few updates stay finite and close to the fp32 run from the same weights and noise stream."""
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    AD = amd.utils.AttrDict
    B = 16
    imgs, nums = synthetic_multi_mnist(B, (50, 50), 2, seed=0)
    x, y = torch.from_numpy(imgs).cuda(), torch.from_numpy(nums).cuda()
    losses = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(0)
        air = amd.mnist_model.AIRonMNIST(x, y, max_steps=3, explore_eps=1e-3, steps_pred_hidden=[128, 64],
                                         transform_var_bias=.5, step_bias=.75, output_multiplier=.5)
        nsp = AD(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3)
        train_step, _ = air.train_step(1e-4, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.), AD(loc=0., scale=1.), nsp,
                                       mfma_dtype=mode)
        assert air._engine.cfg.mfma_dtype == mode
        for _ in range(3):
            train_step()
        losses[mode] = air.rec_loss.item()
        assert np.isfinite(losses[mode]) and torch.isfinite(air._engine.flat_params).all()
    assert abs(losses["bf16"] - losses["f32"]) < 0.05 * abs(losses["f32"]) + 1.0, losses


def test_training_script_counterpart_runs(amd, tmp_path):
    """scripts/multi_mnist.py counterpart: a short run trains, logs the reference's scalar set and checkpoints."""
    from attend_infer_repeat_amd.scripts import multi_mnist
    air = multi_mnist.main(["--iters", "40", "--log-every", "20", "--save-every", "40", "--synthetic-samples", "512",
                            "--eval-batches", "2", "--summary-every", "20", "--results-dir", str(tmp_path)])
    import json, os
    lines = [json.loads(l) for l in open(os.path.join(tmp_path, "multi_mnist", "log.jsonl"))]
    # (the 1000-iteration summaries of the reference's all_summaries: written here every 20 -- round 6's 300 k-update runs died at
    #  their first one, which no test had ever reached)
    summaries = [l for l in lines if l["data"] == "summary"]
    assert summaries and all(l["step"] % 20 == 0 for l in summaries)
    lines = [l for l in lines if l["data"] != "summary"]
    assert {l["data"] for l in lines} == {"train", "test"} and {l["step"] for l in lines} == {0, 20, 40}
    for k in ("loss", "rec_loss", "num_step_acc", "num_step", "prior_loss", "kl_num_steps", "kl_what", "kl_where",
              "baseline_loss", "reinforce_loss", "imp_weight"):
        assert all(np.isfinite(l[k]) for l in lines), k
    assert os.path.exists(os.path.join(tmp_path, "multi_mnist", "model_40.pt"))
    assert int(air.global_step) == 40
    # progress figure (evaluation.py:31-65): inputs / per-step canvases with attention boxes / glimpses
    from attend_infer_repeat_amd.evaluation import make_fig
    air.refresh()
    make_fig(air, str(tmp_path), 40, n_samples=4)
    assert os.path.getsize(os.path.join(tmp_path, "progress_fig_40.png")) > 10_000
    # ... and its CONTENT, panel by panel, from what the engine produced (where / presence / canvases of the last evaluated batch):
    # the attention boxes are those of evaluation.py:23-28 for the engine's `where`, drawn exactly for the steps it marks present
    from test_data import check_progress_figure
    fig = make_fig(air, n_samples=8)
    host = lambda t: t.detach().cpu().numpy()
    pres = host(air.presence)[..., 0]
    assert pres[:, :8].sum() > 0                                    # (step_bias .75: an untrained model takes steps)
    n = check_progress_figure(fig, host(air.obs), host(air.canvas), host(air.glimpse), pres, host(air.where),
                              host(air.num_steps_distrib.prob()[..., 1:]), 8)
    assert n == int(pres[:, :8].sum())


def test_generic_path_decay_rate_and_l2_match_oracle(amd):
    """ops.make_moving_average / decay_rate (model.py:232-239) and l2_weight (model.py:346-353): off in the reference
    script, implemented on the generic path; two consecutive steps so the EMA state matters."""
    ocfg = O.AIRConfig(img_size=(12, 10), crop_size=(5, 4), n_appearance=6, n_hidden=16, inpt_encoder_hidden=(24,),
                       glimpse_encoder_hidden=(20,), glimpse_decoder_hidden=(18,), transform_estimator_hidden=(14,),
                       steps_pred_hidden=(9,), baseline_hidden=(12, 7), max_steps=3, decay_rate=0.8, l2_weight=1e-2)
    B = 9
    params = O.init_params(ocfg, seed=7, bias_std=0.2)
    obs = torch.rand(B, *ocfg.img_size)
    AD = amd.utils.AttrDict
    model = _build_model(amd, ocfg, obs.cuda(), "lstm")
    baseline = amd.modules.BaselineMLP(list(ocfg.baseline_hidden))
    nsp = AD(anneal='exp', init=ocfg.nsp_init, final=ocfg.nsp_final, steps_div=ocfg.nsp_steps_div,
             steps=ocfg.nsp_steps, hold_init=ocfg.nsp_hold_init)
    train_step, _ = model.train_step(ocfg.learning_rate, ocfg.l2_weight, AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                                     AD(loc=0., scale=1.), nsp, baseline=baseline, decay_rate=ocfg.decay_rate)
    _load_oracle_params(amd, model.cell, model.baseline_module, params, "lstm")
    # the EMA variables belong to the model and only a train step updates them (ops.py:46-64: an UPDATE_OP): building the
    # train step and evaluation passes leave them untouched
    assert all(ma.var is None for ma in model._moving_averages.values())
    p64 = {k: v.double() for k, v in params.items()}
    slots = O.rmsprop_init(p64)
    ema_holder = {}
    for it in range(2):
        noise = O.make_noise(ocfg, B, seed=30 + it)
        before = {k: (None if ma.var is None else ma.var.clone()) for k, ma in model._moving_averages.items()}
        model.evaluate(noise={k: v.cuda() for k, v in noise.items()})
        for k, ma in model._moving_averages.items():
            assert (ma.var is None) if before[k] is None else torch.equal(ma.var, before[k]), k
        train_step(noise={k: v.cuda() for k, v in noise.items()})
        n64 = {k: v.double() for k, v in noise.items()}
        n64.update(ema_holder)
        res, grads = O.forward_backward(p64, ocfg, obs.double(), n64, global_step=it)
        ema_holder = {"_ema": n64["_ema"]}
        assert abs(model.reinforce_loss.item() - res["reinforce_loss"].item()) < 2e-4 * (abs(res["reinforce_loss"].item()) + 1)
        assert abs(model.l2_loss.item() - res["l2_loss"].item()) < 1e-4 * (abs(res["l2_loss"].item()) + 1)
        assert abs(model.opt_loss.item() - res["opt_loss"].item()) < 2e-4 * (abs(res["opt_loss"].item()) + 1)
        O.rmsprop_centered_step(p64, grads, slots, ocfg)
    w = model.cell._glimpse_decoder.mlp.layers[1].w
    assert rel(w.data.cpu().double() - params["glimpse_decoder/1/w"].double(),
               p64["glimpse_decoder/1/w"] - params["glimpse_decoder/1/w"].double()) < 5e-3


def test_second_model_does_not_share_moving_averages(amd):
    """Two models in one process keep separate imp_weight EMA state (the reference's variables live in each model's graph)."""
    ocfg = O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3, transform_var_bias=0.5,
                         n_hidden=6)
    AD = amd.utils.AttrDict
    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = _build_model(amd, ocfg, torch.rand(7, *ocfg.img_size).cuda(), "lstm")
        ts, _ = m.train_step(1e-3, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                             AD(anneal=None, init=0.5), baseline=amd.modules.BaselineMLP([6, 4]), decay_rate=0.9)
        models.append((m, ts))
    models[0][1](); models[0][1]()
    assert models[0][0]._moving_averages["imp_weight_moving_var"].var is not None
    assert all(ma.var is None for ma in models[1][0]._moving_averages.values())
    assert models[0][0]._moving_averages is not models[1][0]._moving_averages


def test_where_shift_prior_without_loc_matches_oracle(amd):
    """model.py:203-207: `where_shift_prior` without `loc` centres the shift prior on the posterior's own mean."""
    ocfg = O.AIRConfig(img_size=(12, 10), crop_size=(5, 4), n_appearance=6, n_hidden=16, inpt_encoder_hidden=(24,),
                       glimpse_encoder_hidden=(20,), glimpse_decoder_hidden=(18,), transform_estimator_hidden=(14,),
                       steps_pred_hidden=(9,), baseline_hidden=(12, 7), max_steps=3, where_shift_prior=(None, 0.7))
    B = 9
    params = O.init_params(ocfg, seed=7, bias_std=0.2)
    obs = torch.rand(B, *ocfg.img_size)
    noise = O.make_noise(ocfg, B, seed=8)
    AD = amd.utils.AttrDict
    model = _build_model(amd, ocfg, obs.cuda(), "lstm")
    baseline = amd.modules.BaselineMLP(list(ocfg.baseline_hidden))
    nsp = AD(anneal='exp', init=ocfg.nsp_init, final=ocfg.nsp_final, steps_div=ocfg.nsp_steps_div,
             steps=ocfg.nsp_steps, hold_init=ocfg.nsp_hold_init)
    train_step, _ = model.train_step(ocfg.learning_rate, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                                     AD(scale=0.7), nsp, baseline=baseline)
    _load_oracle_params(amd, model.cell, model.baseline_module, params, "lstm")
    train_step(noise={k: v.cuda() for k, v in noise.items()})
    p64 = {k: v.double() for k, v in params.items()}
    res, grads = O.forward_backward(p64, ocfg, obs.double(), {k: v.double() for k, v in noise.items()}, global_step=0)
    assert abs(model.kl_where.item() - res["kl_where"].item()) < 1e-4 * (abs(res["kl_where"].item()) + 1)
    assert abs(model.opt_loss.item() - res["opt_loss"].item()) < 1e-4 * (abs(res["opt_loss"].item()) + 1)
    for k, p in {"transform/1/w": model.cell._transform_estimator.mlp.layers[1].w,
                 "transform/0/b": model.cell._transform_estimator.mlp.layers[0].b,
                 "lstm/w_gates": model.cell._transition.w_gates}.items():
        assert rel(p.grad, grads[k]) < 1e-3, (k, rel(p.grad, grads[k]))


def _mnist_model(amd, B=8, **kw):
    from attend_infer_repeat_amd.data import synthetic_multi_mnist
    imgs, nums = synthetic_multi_mnist(B, (50, 50), 2, seed=0)
    x, y = torch.from_numpy(imgs).cuda(), torch.from_numpy(nums).cuda()
    return amd.mnist_model.AIRonMNIST(x, y, max_steps=3, explore_eps=1e-3, steps_pred_hidden=[128, 64],
                                      transform_var_bias=.5, step_bias=.75, output_multiplier=.5, **kw)


def test_aironmnist_argument_combinations_the_reference_accepts(amd):
    """model.py:261-265 accepts priors left at None, a shift prior without `loc`, use_reinforce=False and a weighted
    num-steps prior.  The fused engine covers the script's configuration and its plain switches; everything else must fall
    back to the generic path instead of crashing or silently ignoring an argument."""
    AD = amd.utils.AttrDict
    nsp = lambda **kw: AD(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3, **kw)
    N01 = lambda: AD(loc=0., scale=1.)
    # use_reinforce=False: engine path, baseline never built, no REINFORCE term
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-4, 0., N01(), N01(), N01(), nsp(), use_reinforce=False)
    assert air._engine is not None and air._engine.cfg.use_reinforce is False
    ts(); ts()
    assert int(gs) == 2 and np.isfinite(air.opt_loss.item()) and torch.isfinite(air._engine.flat_params).all()
    assert abs(air.opt_loss.item() - air.loss.value.item()) < 1e-6 * abs(air.loss.value.item())
    # priors left at None (the reference's defaults: the KL term is skipped) -> on the engine since round 5
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-4, num_steps_prior=nsp())
    assert air._engine is not None and air._engine.cfg.what_prior is None and air._engine.cfg.where_scale_prior is None
    ts()
    assert int(gs) == 1 and np.isfinite(float(air.opt_loss)) and not hasattr(air, "kl_what")
    assert abs(float(air.prior_loss.value) - float(air.kl_num_steps)) < 1e-5 * (abs(float(air.kl_num_steps)) + 1)
    # shift prior without loc -> on the engine since round 5 (the kernels' NaN-location convention)
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-4, 0., N01(), N01(), AD(scale=1.), nsp())
    assert air._engine is not None and air._engine.cfg.where_shift_prior[0] is None
    ts()
    assert np.isfinite(float(air.kl_where))
    # weighted num-steps KL -> engine, and the weight is honoured
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-4, 0., N01(), N01(), N01(), nsp(weight=3.))
    assert air._engine is not None and air._engine.cfg.nsp_weight == 3.
    ts()
    expect = 3. * float(air.kl_num_steps) + float(air.kl_what) + float(air.kl_where)
    assert abs(float(air.prior_loss.value) - expect) < 1e-4 * (abs(expect) + 1)
    # l2_weight + decay_rate + the RMSProp keyword set (model.py:261-265) -> engine; what stays generic: a custom optimizer class
    # (test below) and a non-MLP baseline
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-4, 1e-3, N01(), N01(), N01(), nsp(), decay_rate=0.9, opt_kwargs=dict(momentum=.5, centered=True, decay=.95))
    eng = air._engine
    assert eng is not None and eng.cfg.l2_weight == 1e-3 and eng.cfg.decay_rate == 0.9 and eng.cfg.rms_momentum == 0.5 and eng.cfg.rms_decay == 0.95
    ts(); ts()
    assert int(gs) == 2 and np.isfinite(air.opt_loss.item()) and torch.isfinite(eng.flat_params).all()
    assert float(air.l2_loss) > 0 and float(air.imp_weight_moving_var) != 1.0
    air = _mnist_model(amd)
    # a non-analytic num-steps prior: sampled step weights, the prior inside the importance weight (model.py:157-163, 339-340)
    ts, gs = air.train_step(1e-4, 0., N01(), N01(), N01(), nsp(analytic=False))
    assert air._engine is not None and air._engine.cfg.nsp_analytic is False
    ts(); ts()
    assert torch.equal(air.prior_step_weight.reshape(-1), air.presence.reshape(-1))
    assert torch.allclose(air.reinforce_imp_weight, air.rec_loss_per_sample + air.prior_loss.per_sample)
    assert int(gs) == 2 and np.isfinite(air.opt_loss.item()) and torch.isfinite(air._engine.flat_params).all()
    # discrete_steps=False (cell.py:150-151): presence = presence_prob, on the engine too
    air = _mnist_model(amd, discrete_steps=False)
    ts, gs = air.train_step(1e-4, 0., N01(), N01(), N01(), nsp())
    assert air._engine is not None and air._engine.cfg.discrete_steps is False
    ts(); ts()
    assert torch.equal(air.presence.reshape(-1), air.presence_prob.reshape(-1))
    assert int(gs) == 2 and np.isfinite(air.opt_loss.item()) and torch.isfinite(air._engine.flat_params).all()


def test_debug_flag_validates_distribution_parameters(amd):
    """cell.py:66-67,130-131,144-145: debug=True turns on validate_args on the three distributions."""
    air = amd.cell.AIRCell((3, 3), (2, 2), 10, debug=True, **make_modules(amd))
    x = torch.rand(4, 3, 3).cuda()
    state = air.initial_state(x)
    air(None, state)                                                     # healthy parameters pass
    with torch.no_grad():
        air._what_distrib.b.fill_(float("nan"))
    with pytest.raises(ValueError, match="what_"):
        air(None, air.initial_state(x))
    quiet = amd.cell.AIRCell((3, 3), (2, 2), 10, debug=False, **make_modules(amd))
    quiet(None, quiet.initial_state(x))
    with torch.no_grad():
        quiet._what_distrib.b.fill_(float("nan"))
    quiet(None, quiet.initial_state(x))                                  # debug=False: no check, like the reference


def test_training_script_resume_is_bit_exact(amd, tmp_path):
    """A run interrupted at a checkpoint and resumed with --resume ends with exactly the parameters of an uninterrupted run
    (parameters, RMSProp slots, step counter, learning rate, Philox state and feeder positions all travel)."""
    from attend_infer_repeat_amd.scripts import multi_mnist
    common = ["--log-every", "1000", "--save-every", "20", "--synthetic-samples", "512", "--eval-batches", "1"]
    a = multi_mnist.main(["--iters", "40", "--results-dir", str(tmp_path / "a")] + common)
    multi_mnist.main(["--iters", "20", "--results-dir", str(tmp_path / "b")] + common)
    b = multi_mnist.main(["--iters", "40", "--results-dir", str(tmp_path / "b"), "--resume",
                          str(tmp_path / "b" / "multi_mnist" / "model_20.pt")] + common)
    a._engine.synchronize(); b._engine.synchronize()
    assert int(b.global_step) == 40
    assert torch.equal(a._engine.flat_params, b._engine.flat_params)
    assert torch.equal(a._engine.flat_mom, b._engine.flat_mom)


def test_custom_optimizer_runs_on_the_generic_path(amd):
    """model.py:265: `optimizer` / `opt_kwargs` are free; anything but the default centred RMSProp is taken as a torch.optim
    class and steps the autograd path (the engine stays out of it)."""
    AD = amd.utils.AttrDict
    air = _mnist_model(amd)
    ts, gs = air.train_step(1e-3, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                            AD(anneal='exp', init=1. - 1e-15, final=1e-7, steps_div=1e4, steps=1e5, hold_init=1e3),
                            optimizer=torch.optim.Adam, opt_kwargs=dict(betas=(0.9, 0.99)))
    assert air._engine is None
    w0 = air.cell._input_encoder.mlp.layers[0].w.detach().clone()
    b0 = air.baseline_module.mlp.layers[0].w.detach().clone()
    ts(); ts()
    assert int(gs) == 2 and np.isfinite(float(air.opt_loss))
    assert not torch.equal(w0, air.cell._input_encoder.mlp.layers[0].w) and not torch.equal(b0, air.baseline_module.mlp.layers[0].w)
    # Adam's first steps move every touched weight by ~lr (model) and ~10 lr (baseline, model.py:363)
    dm = (air.cell._input_encoder.mlp.layers[1].w - 0).abs().max()   # finite
    assert torch.isfinite(dm)
    # the DEFAULT class with other keywords (model.py:265 instantiates RMSPropOptimizer(lr, **opt_kwargs)): momentum=.5 alone means
    # tf.train.RMSPropOptimizer's own defaults for the rest -- decay .9, epsilon 1e-10, NOT centred -- on the HIP update kernel
    air2 = _mnist_model(amd)
    ts2, gs2 = air2.train_step(1e-3, 0., AD(loc=0., scale=1.), AD(loc=0., scale=1.), AD(loc=0., scale=1.),
                               AD(anneal=None, init=0.5), opt_kwargs=dict(momentum=0.5))
    assert air2._engine is not None and air2._custom_optimizer is None          # (round 5: the keyword set runs on the engine)
    assert air2._engine.cfg.rms_centered is False and air2._engine.cfg.rms_momentum == 0.5
    w = air2.cell._input_encoder.mlp.layers[1].w
    w0 = w.detach().clone()
    ts2()
    g = air2._engine.named_grads()["input_encoder/1/w"].double()
    expect = w0.double() - 1e-3 * g / torch.sqrt(0.9 * 1.0 + 0.1 * g * g + 1e-10)        # slots start at ms = 1, mom = 0
    assert torch.allclose(w.detach().double(), expect, rtol=0, atol=2e-7 * float(expect.abs().max()) + 1e-9)
    ts2()
    assert int(gs2) == 2 and torch.isfinite(w).all()
    with pytest.raises(TypeError):
        _mnist_model(amd).train_step(1e-3, num_steps_prior=AD(anneal=None, init=0.5), opt_kwargs=dict(beta1=0.5))

