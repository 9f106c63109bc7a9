"""GPU parity of the fused train step (AIREngine: batched unroll + hand-derived backward + RMSProp, all through the
C ABI) against the CPU oracle's literal per-step restatement with autograd.

Tolerances: the oracle is evaluated in float64 on the same fp32 inputs; the engine computes in fp32 with a different
summation order (batched over T, split-K).  Every tensor is checked two ways: the worst element against the tensor's max
magnitude (fp32 cancellation noise scales with the tensor, not with the element) AND the relative L2 error of the whole
tensor (so small-magnitude elements are not hidden behind one large one).  The bounds are ~10x the worst values measured
on MI355X over all configurations below (profiles/r02_parity_margins.json): outputs 1e-4 / 3e-5, gradients 3e-4 / 2e-4
(typical gradient error is 1e-5..3e-5; the B=17 case is ill-conditioned for this seed -- the ORACLE evaluated in fp32
deviates from its own fp64 evaluation by 8e-4 there, the engine by 1.7e-4)."""
import dataclasses
import json
import os

import numpy as np
import pytest

from attend_infer_repeat_amd.distributed import free_rendezvous_port as D_free_port   # below the ephemeral range
import torch

from oracle import air_oracle as O

pytestmark = pytest.mark.gpu


def make_pair(ocfg: O.AIRConfig, B, seed=1, gstep=20000, bias_std=0.1, mfma_dtype="f32"):
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    fields = {f.name for f in dataclasses.fields(EngineConfig)}
    ecfg = EngineConfig(mfma_dtype=mfma_dtype, **{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields})
    eng = AIREngine(ecfg, B, seed=seed)
    params = O.init_params(ocfg, seed=seed, bias_std=bias_std)
    eng.load_parameters(params)
    obs, _ = O.synthetic_batch(ocfg, B, seed=seed + 10)
    noise = O.make_noise(ocfg, B, seed=seed + 20)
    eng.set_obs(obs.cuda())
    eng.set_noise(noise["eps_where"].cuda(), noise["eps_what"].cuda(), noise["u_pres"].cuda())
    eng.set_global_step(gstep)
    return eng, params, obs, noise


def f64(d):
    return {k: v.double() for k, v in d.items()}


def rel_err(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def l2_err(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


OUT_TOL, OUT_L2 = 1e-4, 3e-5          # per-sample outputs: worst element / tensor max, relative L2
GRAD_TOL, GRAD_L2 = 3e-4, 2e-4        # gradients
_MARGINS = {}


def record_margin(test, name, kind, key, value):
    """AIR_PARITY_MARGINS=<path>: dump the worst observed errors so the tolerances above can be re-derived from a GPU run."""
    path = os.environ.get("AIR_PARITY_MARGINS")
    if not path:
        return
    slot = _MARGINS.setdefault(f"{test}[{name}]", {})
    if value > slot.get(kind, (None, -1.0))[1]:
        slot[kind] = (key, value)
    with open(path, "w") as f:
        json.dump(_MARGINS, f, indent=1, sort_keys=True)


def check_tensor(test, name, kind, key, got, ref, tol, tol_l2):
    e, e2 = rel_err(got, ref), l2_err(got, ref)
    record_margin(test, name, kind + "_max", key, e); record_margin(test, name, kind + "_l2", key, e2)
    if os.environ.get("AIR_PARITY_RECORD_ONLY") == "1":     # measurement run (tools/profile_round.sh): collect, do not judge
        return
    assert e < tol and e2 < tol_l2, (key, e, e2)


CONFIGS = {
    "mnist_b8": (O.AIRConfig(), 8),
    "tiny": (O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3,
                           transform_var_bias=0.5), 10),
    "c4_b4": (O.AIRConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 4),       # BASELINE configs[3] shapes
    "rect_t5": (O.AIRConfig(img_size=(28, 36), crop_size=(9, 12), n_appearance=12, n_hidden=40,
                            inpt_encoder_hidden=(48,), glimpse_encoder_hidden=(33, 21), glimpse_decoder_hidden=(30,),
                            transform_estimator_hidden=(24,), steps_pred_hidden=(16, 8), baseline_hidden=(20,),
                            max_steps=5), 6),
    # edge shapes: a single time step (no BPTT link, dgx aliases dgates[0]) with a batch that is not a multiple of anything;
    # one image; a batch just past a multiple of the 16-row MFMA tile
    "t1_b5": (O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3,
                            transform_var_bias=0.5, max_steps=1), 5),
    "b1": (O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3,
                         transform_var_bias=0.5), 1),
    "mnist_b17": (O.AIRConfig(), 17),
    # a 512-wide first hidden layer at batch 32: the consumer of the K-split obs products then has a LONG K itself (K = 512 >= 8 x
    # min(B, N)); the library must keep it on the A-prologue kernel instead of refusing the launch
    "enc512_b32": (O.AIRConfig(inpt_encoder_hidden=(512, 256)), 32),
    # the sizes the metric is quoted on: BASELINE configs[1] (scripts/multi_mnist.py:24-37: 50x50 / 20x20 / T=3, batch 64) and
    # configs[3] (100x100 canvas, 28x28 glimpse, T=5) at the same batch
    "mnist_b64": (O.AIRConfig(), 64),
    "c4_b64": (O.AIRConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 64),
    # one step at batch 512: 512 rows keep the LATENCY plan while the closing weight-gradient group (K = 512, thousands of tiles) is one
    # the library's wide-tile dispatch takes -- the folded closing update must step aside instead of failing at the first launch
    "t1_b512": (O.AIRConfig(max_steps=1), 512),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_forward_and_gradients_match_oracle(gpu_device, name):
    ocfg, B = CONFIGS[name]
    _forward_and_gradients_match_oracle(name, ocfg, B)


def _random_config(case):
    """a seeded draw of sizes the reference's constructor accepts: image / glimpse shapes (square or not, glimpse sides with every
    residue mod 4), latent and hidden widths that are not multiples of the 16-wide MFMA tile, one to three hidden layers per network,
    1-5 steps, batch sizes around the tile and launch-splitting boundaries"""
    rng = np.random.default_rng(3000 + case)
    def hidden(lo, hi, nmax=3):
        return tuple(int(rng.integers(lo, hi)) for _ in range(int(rng.integers(1, nmax + 1))))
    ocfg = O.AIRConfig(img_size=(int(rng.integers(12, 61)), int(rng.integers(12, 61))),
                       crop_size=(int(rng.integers(4, 25)), int(rng.integers(4, 25))),
                       n_appearance=int(rng.integers(3, 65)), n_hidden=int(rng.choice([24, 40, 64, 100, 256])),
                       inpt_encoder_hidden=hidden(20, 300, 2), glimpse_encoder_hidden=hidden(16, 300, 2),
                       glimpse_decoder_hidden=hidden(16, 300, 2), transform_estimator_hidden=hidden(8, 300, 2),
                       steps_pred_hidden=hidden(8, 130, 2), baseline_hidden=hidden(8, 300, 2),
                       max_steps=int(rng.integers(1, 6)), step_bias=float(rng.uniform(0.0, 1.5)),
                       transform_var_bias=float(rng.uniform(-3.0, 0.5)), output_multiplier=float(rng.uniform(0.3, 1.0)),
                       explore_eps=float(rng.choice([0.0, 1e-3, 0.05])))
    B = int(rng.choice([1, 3, 15, 16, 17, 31, 48, 64, 65, 100, 130]))
    return ocfg, B


@pytest.mark.parametrize("case", range(12))
def test_random_configs_match_oracle(gpu_device, case):
    """the full forward + backward against the float64 oracle on seeded random architectures and batch sizes (plan construction:
    tile / grouping / fusion gates at sizes nobody picked by hand), same bars as the named configurations"""
    ocfg, B = _random_config(case)
    eng, params, obs, noise = make_pair(ocfg, B)
    # (gradient bar: the fixed one, or twice what the oracle itself loses in fp32 on the same inputs -- _check_against_f64_oracle)
    _check_against_f64_oracle("random_configs", "random%d" % case, eng, ocfg, params, obs, noise, out_tol=OUT_TOL)


def _forward_and_gradients_match_oracle(name, ocfg, B):
    eng, params, obs, noise = make_pair(ocfg, B)
    eng.forward(sample_noise=False)
    eng.backward()
    out = eng.outputs()
    res, grads = O.forward_backward(f64(params), ocfg, obs.double(), f64(noise), global_step=20000)
    T = ocfg.max_steps
    # discrete samples must agree exactly for the rest to be comparable
    assert torch.equal(out["presence"].cpu().double(), res["presence"])
    for k in ["what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob", "canvas",
              "final_canvas", "glimpse", "rec_loss_per_sample", "kl_num_steps_per_sample", "kl_what_per_sample",
              "kl_where_per_sample", "num_steps_posterior", "prior_step_weight", "num_steps_log_prob", "baseline"]:
        ref = res[k]
        got = out[k].reshape(ref.shape)
        check_tensor("fwd_bwd", name, "out", k, got, ref, OUT_TOL, OUT_L2)
    for k in ["rec_loss", "kl_num_steps", "kl_what", "kl_where", "loss", "reinforce_loss", "baseline_loss",
              "opt_loss", "imp_weight_mean", "imp_weight_var"]:
        err = abs(out[k].item() - res[k].item()) / (abs(res[k].item()) + 1.0)
        record_margin("fwd_bwd", name, "scalar", k, err)
        assert err <= 2e-5, (k, out[k].item(), res[k].item())
    g = eng.named_grads()
    for k, ref in grads.items():
        check_tensor("fwd_bwd", name, "grad", k, g[k], ref, GRAD_TOL, GRAD_L2)


@pytest.mark.parametrize("name", ["mnist_b8", "rect_t5"])
def test_bf16_mfma_path_matches_bf16_emulating_oracle(gpu_device, name):
    """BASELINE configs[4]: dense products with bf16-rounded operands on the bf16 MFMA, fp32 accumulate.  The oracle
    emulates exactly that arithmetic (O.matmul_mode), so the tolerance stays tight: what is left is summation order plus
    the rare operand that sits within fp32 noise of a bf16 rounding boundary (one bf16 ulp = 2^-8 relative on that
    element).  Outputs 2e-3, gradients 1e-2 of each tensor's max; the plain-fp32 oracle is ~10x further away."""
    ocfg, B = CONFIGS[name]
    eng, params, obs, noise = make_pair(ocfg, B, mfma_dtype="bf16")
    eng.forward(sample_noise=False)
    eng.backward()
    out = eng.outputs()
    with O.matmul_mode("bf16"):
        res, grads = O.forward_backward(f64(params), ocfg, obs.double(), f64(noise), global_step=20000)
    res32, _ = O.forward_backward(f64(params), ocfg, obs.double(), f64(noise), global_step=20000)
    assert torch.equal(out["presence"].cpu().double(), res["presence"])
    for k in ["what", "where", "presence_prob", "final_canvas", "rec_loss_per_sample", "kl_what_per_sample",
              "kl_where_per_sample", "num_steps_posterior", "baseline"]:
        assert rel_err(out[k].reshape(res[k].shape), res[k]) < 2e-3, (k, rel_err(out[k].reshape(res[k].shape), res[k]))
    for k in ["rec_loss", "kl_what", "kl_where", "loss", "opt_loss", "baseline_loss"]:
        assert abs(out[k].item() - res[k].item()) <= 2e-3 * (abs(res[k].item()) + 1.0), (k, out[k].item(), res[k].item())
    g = eng.named_grads()
    bad = {k: rel_err(g[k], ref) for k, ref in grads.items() if not rel_err(g[k], ref) < 1e-2}
    assert not bad, bad
    # the mode is really on: the result differs from exact fp32 by a bf16-sized amount, not by fp32 noise
    d32 = rel_err(out["what"].reshape(res32["what"].shape), res32["what"])
    assert 1e-4 < d32 < 5e-2, d32


def _separate_borderline_draws(noise, res, margin=0.05):
    """Bernoulli draws whose uniform variate sits within `margin` of the presence probability are moved away from it (same
    outcome, but no longer within rounding noise of the threshold).  The presence probabilities depend on the image only
    (cell.py:126-141: the LSTM sees the encoded image and its own state, never the samples), so the outcomes of the oracle
    are unchanged and the bf16-operand engine -- whose probabilities differ by ~1e-3 -- then draws the same ones."""
    u = noise["u_pres"].clone()
    p = res["presence_prob"].reshape(u.shape).to(u.dtype)
    near = (u - p).abs() < margin
    u = torch.where(near & (u < p), (p - margin).clamp(min=0.0), u)
    u = torch.where(near & (u >= p), (p + margin).clamp(max=1.0), u)
    return dict(noise, u_pres=u), int(near.sum().item())


def test_dx_chain_plan_pass_matches_the_per_layer_plan(gpu_device, monkeypatch):
    """Round 6 (measured, not adopted: profiles/r06_dx_chain_rejected.txt): with AIR_DX_CHAIN=1 the throughput plan of the bf16 data path
    replaces runs of per-layer dX launches by row-slab chain launches (air_mlp_dx_chain_bf16).  Same forward (untouched: bit-equal
    outputs), fewer launches, and every gradient agrees with the per-layer plan to the bf16 path's own noise -- the two differ in the
    order of the fp32 accumulation, and each layer consumes bf16 of the previous result."""
    ocfg, B = O.AIRConfig(), 1024
    monkeypatch.setenv("AIR_DX_CHAIN", "0")
    eng0, params, obs, noise = make_pair(ocfg, B, mfma_dtype="bf16")
    monkeypatch.setenv("AIR_DX_CHAIN", "1")
    eng1, *_ = make_pair(ocfg, B, mfma_dtype="bf16")
    n0, n1 = sum(eng0.kernel_launch_count().values()), sum(eng1.kernel_launch_count().values())
    names1 = [n for plan in eng1._single_gpu_step_plans() for _, _, n in plan]
    assert n1 < n0 and names1.count("air_mlp_dx_chain_bf16") >= 3 and eng1._dx_chain_launches == names1.count("air_mlp_dx_chain_bf16")
    for e in (eng0, eng1):
        e.set_noise(noise["eps_where"].cuda(), noise["eps_what"].cuda(), noise["u_pres"].cuda())
        e.forward(sample_noise=False); e.backward(); e.synchronize()
    assert torch.equal(eng0.final_canvas, eng1.final_canvas) and torch.equal(eng0.what, eng1.what)
    g0, g1 = eng0.named_grads(), eng1.named_grads()
    for k in g0:
        a, b = g1[k].double().cpu().reshape(-1), g0[k].double().cpu().reshape(-1)
        assert torch.isfinite(a).all()
        # (bulk: relative L2; tensors that are heavily cancelling batch sums move by more under any change of rounding: bounded loosely)
        assert l2_err(a, b) < 0.2, (k, l2_err(a, b))
    heavy = ["glimpse_decoder/0/w", "glimpse_decoder/1/w", "glimpse_decoder/2/w", "what/w"]
    for k in heavy:
        assert l2_err(g1[k].double().cpu().reshape(-1), g0[k].double().cpu().reshape(-1)) < 2e-2, k


def test_bf16_path_at_batch_1024_matches_bf16_emulating_oracle(gpu_device):
    """BASELINE configs[4] at its own size (batch 1024, 3072 glimpse rows: the throughput-regime plan) against the oracle that
    emulates the bf16-operand arithmetic, evaluated in fp32 (0.6 s on the host).  Outputs AND gradients are compared on every
    image, unconditionally: the (few) Bernoulli draws that sit within bf16 noise of their threshold are moved away from it
    first, for both sides alike, so that no draw can flip."""
    ocfg, B = O.AIRConfig(), 1024
    eng, params, obs, noise = make_pair(ocfg, B, mfma_dtype="bf16")
    with O.matmul_mode("bf16"):
        res0, _ = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    noise, moved = _separate_borderline_draws(noise, res0)
    assert moved < 0.15 * noise["u_pres"].numel(), moved
    eng.set_noise(noise["eps_where"].cuda(), noise["eps_what"].cuda(), noise["u_pres"].cuda())
    eng.forward(sample_noise=False)
    eng.backward()
    out = eng.outputs()
    with O.matmul_mode("bf16"):
        res, grads = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    assert torch.equal(res["presence"], res0["presence"])             # moving the variates changed no outcome
    assert torch.equal(out["presence"].cpu().reshape(ocfg.max_steps, B), res["presence"].reshape(ocfg.max_steps, B))

    def close(k, a, r):
        """bf16 mode at 3072 rows: an operand that sits within fp32 noise of a bf16 rounding boundary flips one bf16 ulp on
        that element, and through `where` that moves a glimpse by a fraction of a pixel -- a handful of canvas pixels next to
        sharp edges then differ by percents.  So the bulk is bounded tightly (relative L2, 99.9th percentile of the absolute
        error against the tensor's max) and the worst single element loosely."""
        a, r = a.double().reshape(-1), r.double().reshape(-1)
        err = (a - r).abs() / (r.abs().max() + 1e-12)
        p999 = torch.quantile(err[:: max(1, err.numel() // 4_000_000)], 0.999).item()
        record_margin("bf16_b1024", "c5", "out_max", k, err.max().item())
        record_margin("bf16_b1024", "c5", "out_p999", k, p999)
        record_margin("bf16_b1024", "c5", "out_l2", k, l2_err(a, r))
        assert l2_err(a, r) < 2e-3 and p999 < 2e-3 and err.max().item() < 0.1, (k, l2_err(a, r), p999, err.max().item())

    for k in ["what", "where", "presence_prob", "final_canvas", "rec_loss_per_sample", "kl_what_per_sample",
              "kl_where_per_sample", "baseline"]:
        close(k, out[k].cpu(), res[k])
    for k in ["rec_loss", "kl_what", "kl_where", "loss", "opt_loss", "baseline_loss"]:
        assert abs(out[k].item() - res[k].item()) <= 2e-3 * (abs(res[k].item()) + 1.0), (k, out[k].item(), res[k].item())
    # Gradients.  Several tensors (input encoder, LSTM, transform MLP: everything that only sees the loss through `where`
    # and the step logits) are batch sums with heavy cancellation: rounding the operands to bf16 moves them by 40-140 % of
    # their norm (bf16-emulating oracle vs the exact-fp32 oracle), so "1 % of the tensor" is not a meaningful bar for
    # them.  The bar that is: the engine must agree with the EMULATION of its arithmetic far better than that arithmetic
    # differs from fp32 -- relative L2 error <= max(5e-3, a quarter of the bf16-vs-fp32 distance); measured 1-14 % of it.
    _, grads32 = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    g = eng.named_grads()
    for k, ref in grads.items():
        spread = l2_err(ref, grads32[k])
        e2, e = l2_err(g[k], ref), rel_err(g[k], ref)
        record_margin("bf16_b1024", "c5", "grad_l2_over_spread", k, e2 / (spread + 1e-30))
        record_margin("bf16_b1024", "c5", "grad_l2", k, e2)
        assert e2 < max(5e-3, 0.25 * spread) and e < 3 * max(5e-3, 0.25 * spread), (k, e, e2, spread)


def test_graph_captured_train_steps_at_batch_64_match_oracle(gpu_device):
    """The thing bench.py times -- hipGraph replays of noise + forward + backward + both RMSProp updates at BASELINE
    configs[1]'s batch 64 -- against O.train_step in float64.  The graph draws its own Philox noise on the device; after each
    replay the noise it used is read back and handed to the oracle, so three consecutive updates are compared."""
    ocfg, B = O.AIRConfig(), 64
    eng, params, obs, noise = make_pair(ocfg, B, gstep=3)
    p64 = f64(params)
    slots = O.rmsprop_init(p64)
    eng.capture()
    prev = {k: v.clone() for k, v in p64.items()}
    for it in range(3):
        eng.train_step()
        eng.synchronize()
        used = {"eps_where": eng.eps_where.cpu().double(), "eps_what": eng.eps_what.cpu().double(),
                "u_pres": eng.u_pres.cpu().double().reshape(ocfg.max_steps, B, 1)}
        used = {k: v.reshape(noise[k].shape) for k, v in used.items()}
        O.train_step(p64, slots, ocfg, obs.double(), used, global_step=3 + it)
        for k, ref in p64.items():
            d_ref = ref - prev[k]
            d_got = eng.params[k].cpu().double() - prev[k]
            record_margin("graph_train", f"step{it}", "delta_max", k, rel_err(d_got, d_ref))
            assert rel_err(d_got, d_ref) < 5e-4, (it, k, rel_err(d_got, d_ref))
        # continue from the ENGINE's parameters so that fp32 rounding of the state does not accumulate into the next comparison
        for k in p64:
            p64[k] = eng.params[k].cpu().double().clone()
            prev[k] = p64[k].clone()
        slots_dev = {"ms": eng.flat_ms, "mg": eng.flat_mg, "mom": eng.flat_mom}
        for sk, flat in slots_dev.items():
            for k in p64:
                o, n = eng.param_offsets[k], eng.param_sizes[k]
                slots[k][sk] = flat[o:o + n].view(eng.param_shapes[k]).cpu().double().clone()
    assert eng.global_step == 6 and eng.step_dev.item() == 6


SWITCHES = {
    # the rest of train_step's arguments (model.py:261-353), each on the fused engine since round 5
    "l2_weight": dict(l2_weight=3e-3),
    "decay_rate": dict(decay_rate=0.9),
    "nsp_weight": dict(nsp_weight=2.5),
    "shift_prior_without_loc": dict(where_shift_prior=(None, 0.7)),
    "rms_kwargs": dict(rms_decay=0.95, rms_momentum=0.5, rms_eps=1e-7),
    "rms_not_centered": dict(rms_centered=False, rms_momentum=0.0),
    "all_together": dict(l2_weight=1e-3, decay_rate=0.8, nsp_weight=0.5, where_shift_prior=(None, 1.3), rms_momentum=0.7),
    # a non-analytic num-steps prior (model.py:157-163, 339-340) and priors left at None (model.py:174, 187)
    "nsp_not_analytic": dict(nsp_analytic=False),
    "what_prior_none": dict(what_prior=None),
    "where_priors_none": dict(where_scale_prior=None, where_shift_prior=None),
    "not_analytic_and_more": dict(nsp_analytic=False, what_prior=None, nsp_weight=2.0, decay_rate=0.9, l2_weight=1e-3),
    # continuous steps (cell.py:150-151): presence = presence_prob, with a gradient through the canvas write (and, under a non-analytic
    # prior, through the step weights of the KL rows)
    "continuous_steps": dict(discrete_steps=False),
    "continuous_not_analytic": dict(discrete_steps=False, nsp_analytic=False, nsp_weight=1.5),
    "continuous_no_reinforce": dict(discrete_steps=False, use_reinforce=False),
    "continuous_not_analytic_no_reinforce": dict(discrete_steps=False, nsp_analytic=False, use_reinforce=False, what_prior=None),
}


@pytest.mark.parametrize("B", [8, 64])
@pytest.mark.parametrize("switch", list(SWITCHES))
def test_train_step_switches_on_the_engine_match_oracle(gpu_device, switch, B):
    """Every further argument of the reference's train_step (model.py:261-353) on the fused, graph-captured engine: L2 on the 2-D
    model variables, EMA-normalised importance weights (the moving averages live on the device and move inside the replayed graph),
    a weighted num-steps prior, a where-shift prior without `loc`, the RMSProp keyword set incl. centered=False.  Three
    hipGraph-replayed updates against O.train_step in float64 fed the noise each replay drew; outputs, losses and the EMA state
    too."""
    ocfg = O.AIRConfig(**SWITCHES[switch])
    eng, params, obs, noise = make_pair(ocfg, B, gstep=3)
    for k, v in SWITCHES[switch].items():
        assert getattr(eng.cfg, k) == v or k == "where_shift_prior"
    assert eng.cfg.where_shift_prior == ocfg.where_shift_prior
    if not ocfg.nsp_analytic and (ocfg.use_reinforce or not ocfg.discrete_steps):
        assert "air_imp_weight" in [e[2] for e in eng._plan_bwd]
    if not ocfg.discrete_steps:
        assert "air_canvas_unroll_bwd_dpresence" in [e[2] for e in eng._plan_bwd]
    p64 = f64(params)
    slots = O.rmsprop_init(p64)
    ema_noise_state = {}
    eng.capture()
    prev = {k: v.clone() for k, v in p64.items()}
    for it in range(3):
        eng.train_step()
        eng.synchronize()
        used = {"eps_where": eng.eps_where.cpu().double(), "eps_what": eng.eps_what.cpu().double(),
                "u_pres": eng.u_pres.cpu().double().reshape(ocfg.max_steps, B, 1)}
        used = {k: v.reshape(noise[k].shape) for k, v in used.items()}
        used.update(ema_noise_state)                       # (the oracle keeps its moving averages in the noise dict: "_ema")
        res, _ = O.train_step(p64, slots, ocfg, obs.double(), used, global_step=3 + it)
        if "_ema" in used:
            ema_noise_state = {"_ema": used["_ema"]}
        out = eng.outputs()
        assert rel_err(out["prior_step_weight"], res["prior_step_weight"]) < 1e-4
        assert rel_err(out["presence"].reshape(-1), res["presence"].reshape(-1)) < 1e-5
        for k in ("opt_loss", "loss", "prior_loss", "kl_where", "kl_what") + (
                ("reinforce_loss", "baseline_loss", "imp_weight_mean", "imp_weight_var") if ocfg.use_reinforce else ()):
            assert abs(out[k].item() - res[k].item()) < 3e-4 * (abs(res[k].item()) + 1.0), (it, k, out[k].item(), res[k].item())
        if ocfg.l2_weight > 0:
            # (a read-out over the CURRENT parameters -- after the update the replay ended with; the L2 gradient inside the step
            #  is what the parameter deltas below check)
            l2_now = ocfg.l2_weight * sum((v * v).sum() / 2 for k, v in p64.items() if v.dim() == 2 and not O.is_baseline_param(k))
            assert abs(out["l2_loss"].item() - l2_now.item()) < 1e-5 * l2_now.item()
        if ocfg.decay_rate is not None:
            assert abs(eng.ema_dev[0].item() - used["_ema"]["mean"].item()) < 1e-4 * (abs(used["_ema"]["mean"].item()) + 1.0)
            assert abs(eng.ema_dev[1].item() - used["_ema"]["var"].item()) < 1e-4 * (abs(used["_ema"]["var"].item()) + 1.0)
        for k, ref in p64.items():
            d_ref, d_got = ref - prev[k], eng.params[k].cpu().double() - prev[k]
            assert rel_err(d_got, d_ref) < 1e-3, (switch, it, k, rel_err(d_got, d_ref))
        for k in p64:
            p64[k] = eng.params[k].cpu().double().clone()
            prev[k] = p64[k].clone()
        for sk, flat in {"ms": eng.flat_ms, "mg": eng.flat_mg, "mom": eng.flat_mom}.items():
            for k in p64:
                o, n = eng.param_offsets[k], eng.param_sizes[k]
                slots[k][sk] = flat[o:o + n].view(eng.param_shapes[k]).cpu().double().clone()
        if "_ema" in ema_noise_state:                      # ... and from the engine's fp32 moving averages
            ema_noise_state["_ema"] = {"mean": eng.ema_dev[0].cpu().double().clone(), "var": eng.ema_dev[1].cpu().double().clone()}
    assert eng.step_dev.item() == 6
    if ocfg.decay_rate is not None:
        # an evaluation pass reads the moving averages without moving them (UPDATE_OPS run with the train step only, ops.py:46-64)
        before = eng.ema_dev.clone()
        eng.forward(); eng.synchronize()
        assert torch.equal(eng.ema_dev, before)
    # gradients of the plain forward / backward against the oracle (the L2 term is part of flat_grads)
    eng2, params2, obs2, noise2 = make_pair(ocfg, B)
    eng2.forward(sample_noise=False); eng2.backward()
    res, ref = O.forward_backward(f64(params2), ocfg, obs2.double(), f64(noise2), global_step=20000)
    grads = eng2.named_grads()
    for k, r in ref.items():
        check_tensor("switches", switch, "grad", k, grads[k], r, GRAD_TOL, GRAD_L2)


def test_bf16_graph_train_step_runs_and_learns(gpu_device):
    ocfg, B = O.AIRConfig(), 32
    eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0, mfma_dtype="bf16")
    eng.forward(); first = eng.outputs()["loss"].item()
    eng.capture()
    for _ in range(200):
        eng.train_step()
    eng.forward(); last = eng.outputs()["loss"].item()
    assert torch.isfinite(eng.flat_params).all() and torch.isfinite(eng.flat_grads).all()
    assert np.isfinite(last) and last < first - 50.0, (first, last)


@pytest.mark.parametrize("variant", ["throughput", "unfused", "mid"])
def test_large_batch_plan_matches_oracle(gpu_device, monkeypatch, variant):
    """B = 704 (T*B = 2112 rows, 44 x 16 LSTM tiles > 512): the plan switches to its throughput variants -- GEMM + pointwise
    LSTM steps, wide-tile GEMM kernels, and every weight gradient deferred to a few all-TN launches at the end of the backward.
    "unfused" additionally forces the separate heads / glimpse-read launches (the path taken when the image is not 16-byte
    addressable or too large for the attend kernels' staging) and the in-chain weight gradients.  Same parity bar as the
    latency-regime plan."""
    if variant == "unfused":
        monkeypatch.setenv("AIR_FUSE_ATTEND_M", "0")
        monkeypatch.setenv("AIR_DEFER_DW_MIN_ROWS", "100000000")
        monkeypatch.setenv("AIR_FUSE_LSTM_WIDE", "0")
    # "mid": B = 272 (816 rows): throughput plan (deferred weight gradients, wide tiles where a launch is large enough) with the
    # 16-wave fused LSTM steps and the prologue riding in the first of them
    ocfg, B = O.AIRConfig(), (272 if variant == "mid" else 704)
    eng, params, obs, noise = make_pair(ocfg, B)
    names = [n for _, _, n in eng._plan_fwd_train + eng._plan_bwd]
    assert ("air_lstm_step_bwd_entry" if variant == "mid" else "air_lstm_pointwise_bwd") in names    # (mid: the BPTT entry folded into its first link)
    if variant == "mid":
        assert eng._defer_dw and "air_lstm_step_fwd_prologue" in names and "air_lstm_step_bwd" in names
        assert eng._plan_bwd_riders is None
    elif variant == "unfused":
        assert "air_attend_fwd" not in names and "air_heads_fwd" in names and not eng._defer_dw
        assert "air_lstm_pointwise_fwd" in names
    else:
        assert "air_attend_fwd" in names and eng._defer_dw
        assert "air_lstm_step_fwd" in names and "air_lstm_pointwise_fwd" not in names       # wide-tile fused LSTM steps
        last = eng._plan_bwd[-1]                                     # ONE launch holds every deferred weight gradient (fp32)
        assert last[2] == "air_gemm_grouped" and last[1][1] > 8 and all(d.ta and not d.tb for d in last[1][0])
    _check_against_f64_oracle("large_batch", variant, eng, ocfg, params, obs, noise)


def _check_against_f64_oracle(test, name, eng, ocfg, params, obs, noise, gstep=20000, out_tol=3e-4):
    """forward + backward of `eng` against the float64 oracle at the latency-regime bar (OUT_L2 / GRAD_TOL / GRAD_L2 above).
    The worst-ELEMENT bound of the outputs is 3e-4 here instead of 1e-4: it is a maximum over 10-100x more elements than at
    batch 64 (2 M canvas pixels at batch 272: measured 1.1e-4 on one edge pixel of one canvas, where a 1e-7 difference in
    `where` moves a bilinear tap weight across a stroke edge), while the relative L2 error -- 5e-6 there -- keeps its bound."""
    eng.forward(sample_noise=False)
    eng.backward()
    out = eng.outputs()
    res, grads = O.forward_backward(f64(params), ocfg, obs.double(), f64(noise), global_step=gstep)
    assert torch.equal(out["presence"].cpu().double(), res["presence"])
    for k in ["what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob", "canvas", "final_canvas",
              "glimpse", "rec_loss_per_sample", "kl_num_steps_per_sample", "kl_what_per_sample", "kl_where_per_sample",
              "num_steps_posterior", "prior_step_weight", "num_steps_log_prob", "baseline"]:
        check_tensor(test, name, "out", k, out[k].reshape(res[k].shape), res[k], out_tol, OUT_L2)
    for k in ["rec_loss", "kl_num_steps", "kl_what", "kl_where", "loss", "reinforce_loss", "baseline_loss", "opt_loss"]:
        err = abs(out[k].item() - res[k].item()) / (abs(res[k].item()) + 1.0)
        record_margin(test, name, "scalar", k, err)
        assert err <= 2e-5, (k, out[k].item(), res[k].item())
    # Gradients: GRAD_TOL / GRAD_L2 -- or, for the few tensors that are batch sums with heavy cancellation at these sizes (the input
    # encoder's first layer: 272-1024 images x T steps through the LSTM), twice what the ORACLE ITSELF loses when it is evaluated
    # in fp32 instead of fp64 on the same inputs: fp32 arithmetic in a different summation order cannot be asked to do better
    # than fp32 arithmetic (measured at batch 272: engine 5.0e-4 / 3.6e-4 on input_encoder/0/w).
    _, grads32 = O.forward_backward(params, ocfg, obs, noise, global_step=gstep)
    g = eng.named_grads()
    for k, ref in grads.items():
        f32_max, f32_l2 = rel_err(grads32[k], ref), l2_err(grads32[k], ref)
        record_margin(test, name, "oracle_f32_vs_f64_max", k, f32_max)
        check_tensor(test, name, "grad", k, g[k], ref, max(GRAD_TOL, 2.0 * f32_max), max(GRAD_L2, 2.0 * f32_l2))


def test_throughput_plan_at_batch_1024_fp32_matches_f64_oracle(gpu_device):
    """BASELINE configs[4]'s batch (1024 images, 3072 glimpse rows) in exact fp32: the throughput plan against the float64
    oracle on every output and every gradient element, same bar as batch 64 (the oracle takes ~6 s on 8 host cores)."""
    ocfg, B = O.AIRConfig(), 1024
    eng, params, obs, noise = make_pair(ocfg, B)
    assert eng._defer_dw
    _check_against_f64_oracle("throughput_b1024", "c2_b1024_f32", eng, ocfg, params, obs, noise)


def test_config4_throughput_plan_matches_f64_oracle(gpu_device):
    """BASELINE configs[3] shapes (100x100 canvas, 28x28 glimpse, T = 5) at batch 416 = 2080 glimpses: the image-major attend
    kernels (exact-T = 5 instantiation), wide-tile GEMMs / LSTM steps and deferred weight gradients against the float64 oracle
    (~11 s on 8 host cores), outputs and all gradients."""
    ocfg, B = O.AIRConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 416
    eng, params, obs, noise = make_pair(ocfg, B)
    assert eng._defer_dw and "air_attend_fwd" in [n for _, _, n in eng._plan_fwd_train]
    _check_against_f64_oracle("throughput_c4", "c4_b416", eng, ocfg, params, obs, noise)


def test_throughput_plan_with_odd_layer_sizes_matches_oracle(gpu_device):
    """The throughput plan on a model none of whose layers fits the wide-tile forms (hidden sizes 48 / 33 / 21 / 30 / 24 / 16 / 8 / 20,
    40 LSTM units, a 28x36 canvas with 9x12 glimpses, T = 5) at batch 300 = 1500 rows: every deferred weight gradient takes the
    16x16-tile launches, the LSTM the 16-wave fused form, and the result still matches the oracle."""
    ocfg, B = CONFIGS["rect_t5"][0], 300
    eng, params, obs, noise = make_pair(ocfg, B)
    assert eng._defer_dw
    _check_against_f64_oracle("throughput_odd", "rect_t5_b300", eng, ocfg, params, obs, noise)
    eng.capture(); eng.train_step(); eng.synchronize()
    assert torch.isfinite(eng.flat_params).all()


def test_throughput_plan_equals_latency_plan_per_sample_on_config4_shapes(gpu_device):
    """BASELINE configs[3] shapes (100x100 canvas, 28x28 glimpse, T = 5) at batch 416 = 2080 glimpses: the throughput plan
    (image-major attend kernels with the exact-T = 5 instantiation, wide-tile GEMMs and LSTM steps) against the SAME images run as
    two batches of 208 on the latency-side kernels (one workgroup per glimpse, 16x16 tiles): every per-sample forward output must
    agree (same arithmetic up to the summation order inside the dense products), and the backward must stay finite."""
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    ocfg = O.AIRConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5)
    B = 416
    fields = {f.name for f in dataclasses.fields(EngineConfig)}
    ecfg = EngineConfig(**{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields})
    params = O.init_params(ocfg, seed=1, bias_std=0.1)
    obs, _ = O.synthetic_batch(ocfg, B, seed=11)
    noise = O.make_noise(ocfg, B, seed=21)

    def run(sel):
        eng = AIREngine(ecfg, len(sel), seed=1)
        eng.load_parameters(params)
        eng.set_obs(obs[sel].cuda())
        eng.set_noise(noise["eps_where"][:, sel].cuda(), noise["eps_what"][:, sel].cuda(), noise["u_pres"][:, sel].cuda())
        eng.set_global_step(20000)
        eng.forward(sample_noise=False); eng.backward()
        out = {k: v.clone() for k, v in eng.outputs().items() if torch.is_tensor(v)}
        assert torch.isfinite(eng.flat_grads).all()
        return eng, out
    idx = torch.arange(B)
    eng_big, big = run(idx)
    assert eng_big._defer_dw and "air_attend_fwd" in [n for _, _, n in eng_big._plan_fwd_train]
    halves = [run(idx[:208])[1], run(idx[208:])[1]]
    assert torch.equal(big["presence"][:, :208], halves[0]["presence"]) and torch.equal(big["presence"][:, 208:], halves[1]["presence"])
    for k, bdim in (("what", 1), ("where", 1), ("presence_prob", 1), ("final_canvas", 0), ("rec_loss_per_sample", 0),
                    ("kl_what_per_sample", 0), ("glimpse", 1)):
        if k not in big:
            continue
        ref = torch.cat([halves[0][k], halves[1][k]], dim=bdim)
        assert rel_err(big[k], ref) < 2e-4, (k, rel_err(big[k], ref))


@pytest.mark.parametrize("B,mfma", [(64, "f32"), (1024, "f32"), (1024, "bf16")])
def test_batch_permutation_equivariance_at_full_size(gpu_device, B, mfma):
    """Size-independent property at BASELINE sizes (configs[1]: B=64; configs[4]: B=1024, where the oracle is too slow to be
    the checker): images are independent through the whole forward, so permuting the batch (images and their noise) must
    permute every per-sample output EXACTLY -- each output row / canvas is accumulated in the same order wherever its tile
    sits -- and leave the batch-summed gradients unchanged up to summation order."""
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    ocfg = O.AIRConfig()
    obs, _ = O.synthetic_batch(ocfg, B, seed=3)
    noise = O.make_noise(ocfg, B, seed=4)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    params = O.init_params(ocfg, seed=1, bias_std=0.1)
    outs, grads = [], []
    for pm in (None, perm):
        eng = AIREngine(EngineConfig(mfma_dtype=mfma), B, seed=1)
        eng.load_parameters(params)
        o, nz = (obs, noise) if pm is None else (obs[pm], {k: v[:, pm] for k, v in noise.items()})
        eng.set_obs(o.cuda())
        eng.set_noise(nz["eps_where"].cuda(), nz["eps_what"].cuda(), nz["u_pres"].cuda())
        eng.set_global_step(20000)
        eng.forward(sample_noise=False); eng.backward()
        outs.append({k: v.clone() for k, v in eng.outputs().items() if torch.is_tensor(v)})
        grads.append({k: v.clone() for k, v in eng.named_grads().items()})
    a, b = outs
    pc = perm.cuda()
    for k in ["what", "where", "presence", "presence_prob", "glimpse"]:                 # [T, B, ...]
        assert torch.equal(a[k].reshape(ocfg.max_steps, B, -1)[:, pc], b[k].reshape(ocfg.max_steps, B, -1)), k
    for k in ["final_canvas", "rec_loss_per_sample", "kl_what_per_sample", "kl_where_per_sample", "kl_num_steps_per_sample",
              "num_steps_log_prob", "baseline"]:                                        # [B, ...]
        assert torch.equal(a[k].reshape(B, -1)[pc], b[k].reshape(B, -1)), k
    for k in grads[0]:
        assert rel_err(grads[1][k], grads[0][k]) < 2e-4, (k, rel_err(grads[1][k], grads[0][k]))


def test_train_step_updates_match_oracle(gpu_device):
    ocfg, B = CONFIGS["mnist_b8"]
    eng, params, obs, noise = make_pair(ocfg, B, gstep=3)
    p64 = f64(params)
    slots = O.rmsprop_init(p64)
    cfg64 = ocfg
    for it in range(2):
        eng.forward(sample_noise=False); eng.backward(); eng.optimizer_step()
        O.train_step(p64, slots, cfg64, obs.double(), f64(noise), global_step=3 + it)
    eng.synchronize()
    for k, ref in p64.items():
        delta_ref = ref - params[k].double()
        delta = eng.params[k].cpu().double() - params[k].double()
        assert rel_err(delta, delta_ref) < 5e-3, (k, rel_err(delta, delta_ref))
    assert eng.global_step == 5 and eng.step_dev.item() == 5


@pytest.mark.parametrize("case", [0, 3, 5, 8])
def test_train_step_updates_match_oracle_on_random_configs(gpu_device, case):
    """two full updates (forward, backward, both centred-RMSProp optimisers, step counter) on seeded random architectures against the
    float64 oracle's train step: the parameter DELTAS agree (same bar as the named configuration)"""
    ocfg, B = _random_config(case)
    eng, params, obs, noise = make_pair(ocfg, B, gstep=3)
    p64 = f64(params)
    slots = O.rmsprop_init(p64)
    for it in range(2):
        eng.forward(sample_noise=False); eng.backward(); eng.optimizer_step()
        O.train_step(p64, slots, ocfg, obs.double(), f64(noise), global_step=3 + it)
    eng.synchronize()
    for k, ref in p64.items():
        delta_ref = ref - params[k].double()
        delta = eng.params[k].cpu().double() - params[k].double()
        assert rel_err(delta, delta_ref) < 5e-3, (k, rel_err(delta, delta_ref))
    assert eng.global_step == 5 and eng.step_dev.item() == 5


@pytest.mark.parametrize("which", ["mnist_b8", 1, 4, 7, 10])
def test_graph_replay_equals_eager(gpu_device, which):
    """hipGraph-captured step == eager step from the same state and the same Philox counter (the named configuration and four
    seeded random architectures: whatever launches the plan is made of, capture must not change a bit)."""
    ocfg, B = CONFIGS[which] if isinstance(which, str) else _random_config(which)
    eng_a, params, obs, noise = make_pair(ocfg, B, seed=3, gstep=0)
    eng_b, _, _, _ = make_pair(ocfg, B, seed=3, gstep=0)
    eng_b.capture()
    for _ in range(3):
        eng_a.train_step()
        eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    assert torch.equal(eng_a.noise_normal, eng_b.noise_normal)            # same counter-based noise stream
    assert eng_a.step_dev.item() == eng_b.step_dev.item() == 3
    # every kernel reduces in a fixed order (no float atomics): replay is bitwise identical to eager
    assert torch.equal(eng_b.flat_grads, eng_a.flat_grads)
    assert torch.equal(eng_b.flat_params, eng_a.flat_params)


@pytest.mark.parametrize("name", ["mnist_b8", "tiny"])
def test_optimizer_riders_equal_closing_update(gpu_device, monkeypatch, name):
    """Single-GPU latency-regime steps update every parameter whose gradient is final before the BPTT chain from extra
    workgroups of the BPTT launches (air_lstm_pointwise_bwd_opt / air_lstm_step_bwd_opt) and only the head of the flat buffer
    in the closing launch: bit-identical parameters and RMSProp slots to the single closing air_step_epilogue."""
    ocfg, B = CONFIGS[name]
    monkeypatch.setenv("AIR_OPT_FOLD", "0")                      # (the closing launch itself: its fold has a test of its own below)
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert eng_a._plan_bwd_riders is not None and any(n.endswith("_opt") for _, _, n in eng_a._plan_bwd_riders)
    covered = sorted((s.lo, s.hi) for s in eng_a._rider_slices)
    assert covered[-1][1] == eng_a.n_total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert eng_a._plan_opt_rest[0][1][6].value == covered[0][0]           # the closing launch ends where the riders start
    monkeypatch.setenv("AIR_OPT_RIDERS", "0")
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert eng_b._plan_bwd_riders is None
    eng_a.capture(); eng_b.capture()
    for _ in range(3):
        eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom", "flat_grads"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k
    assert eng_a.step_dev.item() == eng_b.step_dev.item() == 3


@pytest.mark.parametrize("name", ["mnist_b8", "mnist_b64", "c4_b64", "enc512_b32", "tiny", "rect_t5", "t1_b5", "t1_b512"])
def test_folded_closing_update_equals_closing_launch(gpu_device, monkeypatch, name):
    """Round 5: the closing air_step_epilogue of the single-GPU latency-regime step is folded into the LAST
    backward launch -- its weight-gradient tiles apply centred RMSProp to the elements they finish (air_gemm_grouped_opt), rider
    workgroups update what earlier launches left final and advance the counters.  Bit-identical parameters, slots, gradients,
    step counter and Philox offset to the plan with the closing launch and to the plan without any rider, over several
    graph-replayed updates; one launch fewer; layouts the fold does not understand keep their closing launch."""
    ocfg, B = CONFIGS[name]
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    folded = eng_a._fold is not None
    head_whole = all(eng_a.param_sizes[k] % 4 == 0 for k in eng_a.param_shapes if eng_a.param_offsets[k] < eng_a.param_offsets["transform/0/w"])
    if name in ("mnist_b8", "mnist_b64", "c4_b64", "enc512_b32"):
        assert folded, "the standard architectures must take the folded plan"
    if folded:
        assert head_whole and eng_a._plan_opt_rest == [] and eng_a._plan_bwd_riders[-1][2] == "air_gemm_grouped_opt"
        assert sum(eng_a.kernel_launch_count().values()) == len(eng_a._plan_fwd_train) + len(eng_a._plan_bwd)
        # the fold and the rider ranges cover the head of the flat buffers exactly once
        f = eng_a._fold
        ranges = [(f.range_lo[j], f.range_hi[j]) for j in range(f.n_ranges)]
        g0 = eng_a.flat_grads.data_ptr()
        folds = [(f, eng_a._plan_bwd_riders[-1])]
        if getattr(eng_a, "_fold_early", None) is not None:
            # (round 5, late: the LSTM's weight gradients and their update one launch earlier; no rider slices, no counters there)
            fe = eng_a._fold_early
            assert eng_a._plan_bwd_riders[-2][2] == "air_gemm_grouped_opt" and fe.n_ranges == 0 and not fe.global_step_dev
            lw = (eng_a.grads["lstm/w_gates"].data_ptr() - g0) // 4
            folds.append((fe, eng_a._plan_bwd_riders[-2]))
        if name in ("mnist_b8", "mnist_b64", "c4_b64"):
            assert getattr(eng_a, "_fold_early", None) is not None, "the standard architectures fold the LSTM's update one launch early"
        for ff, entry in folds:
            arr, n = entry[1][:2]
            for i in range(n):
                if (ff.fold_mask >> i) & 1:
                    off = (arr[i].C - g0) // 4
                    ranges.append((off, off + arr[i].M * arr[i].N))
                    if arr[i].colsum:
                        ranges.append(((arr[i].colsum - g0) // 4, (arr[i].colsum - g0) // 4 + arr[i].N))
        ranges.sort()
        assert ranges[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert ranges[-1][1] == min(s.lo for s in eng_a._rider_slices)
    monkeypatch.setenv("AIR_OPT_FOLD", "0")
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert eng_b._fold is None and (eng_b._plan_opt_rest is None or len(eng_b._plan_opt_rest) == 1)
    monkeypatch.setenv("AIR_OPT_RIDERS", "0")
    eng_c, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert eng_c._plan_bwd_riders is None
    for e in (eng_a, eng_b, eng_c):
        e.capture()
    for _ in range(4):
        for e in (eng_a, eng_b, eng_c):
            e.train_step()
    for e in (eng_a, eng_b, eng_c):
        e.synchronize()
    for other in (eng_b, eng_c):
        for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom", "flat_grads", "rng_state", "step_dev"):
            assert torch.equal(getattr(eng_a, k), getattr(other, k)), k
    assert eng_a.step_dev.item() == 4
    # eager launches of the folded plan give the same bits as its replay
    eng_a.release_graphs()
    eng_b.release_graphs()
    eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k


@pytest.mark.parametrize("name", ["mnist_b8", "mnist_b64", "c4_b64", "enc512_b32", "tiny", "rect_t5", "t1_b5", "mnist_b17"])
def test_first_step_with_folded_input_product_equals_two_launches(gpu_device, monkeypatch, name):
    """Round 5: in the latency regime the hoisted product gx = enc_out . W_x + b rides in the FIRST LSTM step
    (air_lstm_first_step_fwd: step 0's recurrent operand is the one-row initial state, so the tile accumulates both contractions
    side by side) instead of a launch of its own: one dependent launch fewer, and -- same sums in the same order -- every buffer of
    the step bit-identical to the plan with the separate gx launch (AIR_FOLD_GX=0), over graph-replayed updates."""
    ocfg, B = CONFIGS[name]
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    monkeypatch.setenv("AIR_FOLD_GX", "0")
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert not eng_b._fold_gx
    if len(ocfg.inpt_encoder_hidden) > 1:
        assert eng_a._fold_gx and len(eng_a._plan_fwd_train) == len(eng_b._plan_fwd_train) - 1
        assert any(n == "air_lstm_first_step_fwd" for _, _, n in eng_a._plan_fwd_train)
    else:
        assert not eng_a._fold_gx                            # (a single encoder layer: its K-split halves are reduced by the gx product)
    for e in (eng_a, eng_b):
        e.forward(sample_noise=False); e.backward(); e.synchronize()
    for k in ("gx", "gate_act", "h_seq", "c_seq", "flat_grads", "final_canvas", "kl_what_row"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k
    eng_a.capture(); eng_b.capture()
    for _ in range(3):
        eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom", "flat_grads", "noise_normal", "h_seq"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k


@pytest.mark.parametrize("name", ["mnist_b8", "mnist_b64", "c4_b64", "tiny", "rect_t5", "t1_b5", "mnist_b17", "b1"])
def test_what_head_in_one_launch_equals_product_plus_sampling(gpu_device, monkeypatch, name):
    """Round 5: the `what` head of a latency-regime step -- q = ge_out . W + b, the reparameterised sample with its KL terms, the
    latent columns of the baseline input -- is one launch (air_what_head_fwd) instead of a GEMM launch + air_what_sample_pack.
    q, loc, scale, the sample and the baseline input are bit-identical to the two launches (same K split, same order); the KL rows
    are the tiles' shares added in tile order instead of one 64-lane tree, so they -- and what depends on them -- agree to rounding.
    A forward() on its own still leaves complete KL rows; three graph-replayed updates end within rounding of each other."""
    ocfg, B = CONFIGS[name]
    eng_a, params, obs, noise = make_pair(ocfg, B, seed=3, gstep=0)
    monkeypatch.setenv("AIR_FUSE_WHAT_HEAD", "0")
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert eng_a._what_head and not eng_b._what_head
    assert len(eng_a._plan_fwd_train) == len(eng_b._plan_fwd_train) - 1
    for e in (eng_a, eng_b):
        e.forward(sample_noise=False); e.synchronize()
    for k in ("q", "what", "what_loc", "what_scale", "base_lat"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k
    assert l2_err(eng_a.kl_what_row, eng_b.kl_what_row) < 1e-6          # complete after forward() alone
    oa, ob = eng_a.outputs(), eng_b.outputs()
    for k in ("kl_what", "loss", "opt_loss", "final_canvas", "reinforce_loss"):
        assert rel_err(oa[k], ob[k]) < 1e-6, k
    for e in (eng_a, eng_b):
        e.backward(); e.synchronize()
    assert l2_err(eng_a.kl_what_row, eng_b.kl_what_row) < 1e-6          # ... and after the backward re-added the shares
    for k in eng_a.grads:
        assert l2_err(eng_a.grads[k], eng_b.grads[k]) < 2e-6, k
    res, ref = O.forward_backward(f64(params), ocfg, obs.double(), f64(noise), global_step=0)
    check_tensor("what_head", name, "out", "kl_what_per_sample", oa["kl_what_per_sample"], res["kl_what_per_sample"], OUT_TOL, OUT_L2)
    eng_a.capture(); eng_b.capture()
    for _ in range(3):
        eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    assert torch.equal(eng_a.noise_normal, eng_b.noise_normal)
    assert l2_err(eng_a.flat_params, eng_b.flat_params) < 1e-5
    assert l2_err(eng_a.kl_what_row, eng_b.kl_what_row) < 1e-4          # (inside the train step only the backward adds the shares)


@pytest.mark.parametrize("name", ["mnist_b8", "mnist_b64", "tiny", "rect_t5", "t1_b5", "mnist_b17", "b1"])
def test_what_head_backward_in_the_decoder_epilogue_equals_its_own_launch(gpu_device, monkeypatch, name):
    """Round 5: the backward of the `what` head (air_gauss_sample_bwd_nvil: d q from d what, the KL weights and the stored loc / scale,
    with NVIL and the KL-share sum riding) is folded into the launch that forms d what -- the decoder's first-layer dX writes dq from
    its epilogue, NVIL and the share sum ride behind its tiles (air_gemm_grouped_gauss_bwd).  One element function serves both forms:
    every gradient, the NVIL scalars and the KL rows are BIT-identical to the plan with the separate launch, eagerly and over
    graph-replayed updates; one launch fewer."""
    ocfg, B = CONFIGS[name]
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    monkeypatch.setenv("AIR_FUSE_GAUSS_BWD", "0")
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    assert not eng_b._fold_gauss_bwd
    if name.startswith("mnist"):
        assert eng_a._fold_gauss_bwd
    if eng_a._fold_gauss_bwd:
        assert len(eng_a._plan_bwd) == len(eng_b._plan_bwd) - 1
        assert not any(n.startswith("air_gauss_sample_bwd") for _, _, n in eng_a._plan_bwd)
    # (plans whose last decoder launch also carries the baseline's first-layer weight gradients -- no fused canvas launch -- or that
    #  exceed the 16x16-tile dispatch keep the launch of their own: the comparison below then checks two identical plans)
    for e in (eng_a, eng_b):
        e.forward(sample_noise=False); e.backward(); e.synchronize()
    for k in ("dq", "flat_grads", "nvil_out", "dlogp", "dbase", "kl_what_row", "rec"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k
    eng_a.capture(); eng_b.capture()
    for _ in range(3):
        eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom", "flat_grads", "nvil_out", "kl_what_row"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k


@pytest.mark.parametrize("name", ["mnist_b8", "t1_b5"])
def test_multi_step_replay_equals_single_steps(gpu_device, name):
    """capture(steps_per_replay=K): K consecutive updates in one graph replay, step j reading its batch from slot j of the
    observation ring -- bit-identical to K single-step replays fed the same batches (device-side step counter, Philox offset and
    annealed prior advance inside the graph)."""
    ocfg, B = CONFIGS[name]
    K = 3
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    batches = [O.synthetic_batch(ocfg, B, seed=20 + j)[0].cuda() for j in range(2 * K)]
    eng_a.capture(); eng_b.capture(steps_per_replay=K)
    for r in range(2):
        for j in range(K):
            eng_a.train_step(batches[r * K + j])
            eng_b.set_obs_slot(j, batches[r * K + j])
        eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    assert eng_a.global_step == eng_b.global_step == 2 * K and eng_b.step_dev.item() == 2 * K
    for k in ("flat_params", "flat_ms", "flat_mg", "flat_mom", "flat_grads", "noise_normal"):
        assert torch.equal(getattr(eng_a, k), getattr(eng_b, k)), k
    # the engine's own (single-step) plans still work on the ordinary buffer afterwards
    eng_b.release_graphs(); eng_a.release_graphs()
    eng_a.train_step(batches[0]); eng_b.train_step(batches[0])
    eng_a.synchronize(); eng_b.synchronize()
    assert torch.equal(eng_a.flat_params, eng_b.flat_params)


def test_device_feeder_draws_the_batch_inside_the_step(gpu_device):
    """attach_dataset: the first launch of the (captured) train step gathers the batch from an HBM-resident dataset, indices
    from Philox(seed, device step counter) or sequential; the step then equals a step fed that batch by hand."""
    ocfg, B = CONFIGS["mnist_b8"]
    P = ocfg.img_size[0] * ocfg.img_size[1]
    N = 37
    data = torch.stack([O.synthetic_batch(ocfg, 1, seed=300 + i)[0][0] for i in range(N)]).reshape(N, P).cuda()
    eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    eng_a.attach_dataset(data, shuffle=True, seed=11)
    assert eng_a._plan_fwd_train[0][2] in ("air_batch_gather", "air_gemm_grouped_gather")     # (folded into the first product where it fits)
    eng_a.capture(); eng_b.capture()
    seen = []
    for step in range(4):
        eng_a.train_step(); eng_a.synchronize()
        idx = eng_a.batch_idx.clone()
        assert int(idx.min()) >= 0 and int(idx.max()) < N
        assert torch.equal(eng_a.obs, data[idx])
        seen.append(idx.cpu())
        eng_b.train_step(data[idx]); eng_b.synchronize()
        assert torch.equal(eng_a.flat_params, eng_b.flat_params), step
    assert not torch.equal(seen[0], seen[1])                       # a new draw every step
    eng_c, *_ = make_pair(ocfg, B, seed=3, gstep=0)                # same seed, same step counter -> same indices
    eng_c.attach_dataset(data, shuffle=True, seed=11); eng_c.train_step(); eng_c.synchronize()
    assert torch.equal(eng_c.batch_idx.cpu(), seen[0])
    eng_c.attach_dataset(data, shuffle=False)                      # sequential: (step*B + b) mod N, step = 1 by now
    eng_c.train_step(); eng_c.synchronize()
    assert torch.equal(eng_c.batch_idx.cpu(), (torch.arange(B) + 1 * B) % N)
    # several updates per replay: every step of the replay draws its own batch
    eng_d, *_ = make_pair(ocfg, B, seed=3, gstep=0)
    eng_d.attach_dataset(data, shuffle=True, seed=11); eng_d.capture(steps_per_replay=2)
    eng_d.train_step(); eng_d.train_step(); eng_d.synchronize()
    assert torch.equal(eng_d.flat_params, eng_a.flat_params)


@pytest.mark.parametrize("name", ["mnist_b64", "c4_b64", "mnist_b8"])
def test_feeder_gather_folded_into_the_first_product_equals_the_gather_launch(gpu_device, monkeypatch, name):
    """Round 6: with a dataset attached the latency-regime step opens with the products over the pixels of obs; the feeder's gather is
    folded into their A-operand load (air_gemm_grouped_gather: rows read from the dataset through the Philox index, written to `obs` by
    the first column of tiles) instead of being a launch of its own.  Same indices, same observation buffer, and BIT-identical
    parameters / optimiser state over graph-replayed updates (the product's arithmetic is untouched); one launch fewer."""
    ocfg, B = CONFIGS[name]
    P = ocfg.img_size[0] * ocfg.img_size[1]
    N = 53
    data = torch.stack([O.synthetic_batch(ocfg, 1, seed=500 + i)[0][0] for i in range(N)]).reshape(N, P).cuda()
    engs = []
    for fold in ("0", "1"):
        monkeypatch.setenv("AIR_FOLD_GATHER", fold)
        e, *_ = make_pair(ocfg, B, seed=3, gstep=0)
        e.attach_dataset(data, shuffle=True, seed=5)
        engs.append(e)
    e0, e1 = engs
    assert e0._plan_fwd_train[0][2] == "air_batch_gather" and not e0._fold_gather
    assert e1._plan_fwd_train[0][2] == "air_gemm_grouped_gather" and e1._fold_gather
    assert sum(e1.kernel_launch_count().values()) == sum(e0.kernel_launch_count().values()) - 1
    for e in engs:
        e.capture()
    for step in range(4):
        for e in engs:
            e.train_step(); e.synchronize()
        assert torch.equal(e0.batch_idx, e1.batch_idx) and torch.equal(e1.obs, data[e1.batch_idx]), step
        assert torch.equal(e0.flat_params, e1.flat_params) and torch.equal(e0.flat_mom, e1.flat_mom), step
    # the sequential walk as well
    for e in engs:
        e.attach_dataset(data, shuffle=False); e.train_step(); e.synchronize()
    assert torch.equal(e0.batch_idx, e1.batch_idx) and torch.equal(e0.flat_params, e1.flat_params)


def test_feeder_gather_in_the_bf16_prologue_equals_the_gather_launch(gpu_device, monkeypatch):
    """Round 6, bf16 data path (BASELINE configs[4], batch 1024): the step opens with the prologue whose extra workgroups write the bf16
    mirror of obs; with a dataset attached they read the rows from the dataset through the Philox index and write obs AND the mirror
    (air_step_prologue_gather_cvt) -- no gather launch.  Same indices, same batch, same mirror, bit-identical updates; one launch fewer."""
    ocfg, B = O.AIRConfig(), 1024
    P = ocfg.img_size[0] * ocfg.img_size[1]
    N = 97
    data = torch.stack([O.synthetic_batch(ocfg, 1, seed=700 + i)[0][0] for i in range(N)]).reshape(N, P).cuda()
    engs = []
    for fold in ("0", "1"):
        monkeypatch.setenv("AIR_FOLD_GATHER", fold)
        e, *_ = make_pair(ocfg, B, seed=3, gstep=0, mfma_dtype="bf16")
        e.attach_dataset(data, shuffle=True, seed=5)
        engs.append(e)
    e0, e1 = engs
    assert [n for _, _, n in e0._plan_fwd_train[:2]] == ["air_batch_gather", "air_step_prologue_cvt"] and not e0._fold_gather
    assert e1._plan_fwd_train[0][2] == "air_step_prologue_gather_cvt" and e1._fold_gather
    assert sum(e1.kernel_launch_count().values()) == sum(e0.kernel_launch_count().values()) - 1
    for e in engs:
        e.capture()
    for step in range(3):
        for e in engs:
            e.train_step(); e.synchronize()
        assert torch.equal(e0.batch_idx, e1.batch_idx) and torch.equal(e1.obs.reshape(B, P), data[e1.batch_idx]), step
        assert torch.equal(e0.obs16, e1.obs16) and torch.equal(e1.obs16.reshape(B, P), data[e1.batch_idx].to(torch.bfloat16)), step
        assert torch.equal(e0.flat_params, e1.flat_params) and torch.equal(e0.flat_mom, e1.flat_mom), step
    for e in engs:                                   # the sequential walk as well
        e.attach_dataset(data, shuffle=False); e.train_step(); e.synchronize()
    assert torch.equal(e0.batch_idx, e1.batch_idx) and torch.equal(e0.flat_params, e1.flat_params)


def test_noise_changes_every_step_and_prior_anneals(gpu_device):
    ocfg, B = CONFIGS["tiny"]
    eng, *_ = make_pair(ocfg, B, gstep=0)
    eng.capture()
    eng.train_step(); eng.synchronize()
    n1 = eng.noise_normal.clone(); p1 = eng.prior_dev.clone()
    eng.set_global_step(50000)
    eng.train_step(); eng.synchronize()
    assert not torch.equal(n1, eng.noise_normal)
    s = O.steps_prior_success_prob(ocfg, 50000)
    ref = O.geometric_prior(s, ocfg.max_steps)
    assert torch.allclose(eng.prior_dev.cpu(), ref, rtol=1e-12, atol=0)
    assert not torch.equal(p1, eng.prior_dev)


def test_loss_decreases_on_fixed_batch(gpu_device):
    """Sanity of the whole loop at the reference script's learning rate (multi_mnist.py:24): 200 graph-replayed updates
    on one batch reduce the ELBO loss and stay finite.  (At 10x that rate the raw where-scale runs to -50, softplus
    underflows and KL(where) = +inf -- in the reference's arithmetic as well; AIR is known to be touchy, README.md:37.)"""
    ocfg, B = O.AIRConfig(), 32
    eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0)
    eng.forward(); first = eng.outputs()["loss"].item()
    eng.capture()
    for _ in range(200):
        eng.train_step()
    eng.forward(); last = eng.outputs()["loss"].item()
    assert torch.isfinite(eng.flat_params).all() and torch.isfinite(eng.flat_grads).all()
    assert np.isfinite(last) and last < first - 50.0, (first, last)


def test_data_parallel_path_on_one_gpu_rccl(gpu_device):
    """The multi-GPU step structures run with a 1-rank RCCL group; each must equal the single-GPU graph bit for bit:
      * the collective CAPTURED in the step's graph (air_allreduce_sum on the engine stream: still one graph replay per step),
        plain and with the tail slice all-reduced on a forked captured stream;
      * the fallback: graph A (fwd+bwd) -> torch.distributed all-reduce -> graph B (RMSProp)."""
    import ctypes
    import os
    import socket
    import torch.distributed as dist
    from attend_infer_repeat_amd import distributed as D
    from attend_infer_repeat_amd import hip as H
    port = D_free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ocfg, B = CONFIGS["mnist_b8"]
        eng_a, *_ = make_pair(ocfg, B, seed=3, gstep=0)
        eng_a.capture()
        for _ in range(3):
            eng_a.train_step()
        eng_a.synchronize()
        # --- captured RCCL collective (opt-in protocol of DataParallelEngine: AIR_DP_COLLECTIVE=rccl-captured) ----------
        comm = D.create_rccl_comm(torch.device("cuda", 0))
        comm_side = D.create_rccl_comm(torch.device("cuda", 0))          # the forked stream reduces on a communicator of its own
        assert D.comm_count(comm) == 1 and D.comm_count(comm_side) == 1   # ncclCommCount through the C ABI
        assert D.selftest_comm(comm, torch.device("cuda", 0), eng_a.stream)
        for overlap in (False, True):
            eng_c, *_ = make_pair(ocfg, B, seed=3, gstep=0)
            eng_c.world_size = 1
            eng_c.capture(comm=comm, overlap=overlap, comm_side=comm_side if overlap else None)
            assert eng_c._graph is not None and eng_c._graph_opt is None and eng_c._graph_has_opt      # ONE graph
            for _ in range(3):
                eng_c.train_step()
            eng_c.synchronize()
            assert torch.equal(eng_a.flat_params, eng_c.flat_params), overlap
            assert torch.equal(eng_a.flat_mom, eng_c.flat_mom), overlap
            eng_c.release_graphs()
        # the stand-alone entry really sums: with one rank it must leave the buffer untouched
        g = torch.arange(1000, dtype=torch.float32, device="cuda")
        st = H.lib().air_allreduce_sum(H._p(g), ctypes.c_size_t(g.numel()), comm, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert st == 0 and torch.equal(g, torch.arange(1000, dtype=torch.float32, device="cuda"))
        assert H.lib().air_comm_destroy(comm_side) == 0 and H.lib().air_comm_destroy(comm) == 0
        # --- the wrapper: world 1 -> plain single graph, collective "none" ---------------------------------------------
        eng_d, *_ = make_pair(ocfg, B, seed=3, gstep=0)
        dp = D.DataParallelEngine(eng_d)
        assert dp.collective == "none" and dp.world == 1
        for _ in range(3):
            dp.train_step()
        eng_d.synchronize()
        assert torch.equal(eng_a.flat_params, eng_d.flat_params)
        # --- fallback: two graphs with a host-issued collective in between ------------------------------------------------
        eng_b, *_ = make_pair(ocfg, B, seed=3, gstep=0)
        eng_b.world_size = 1
        eng_b.capture(split_optimizer=True)
        calls = []

        def allreduce(gr):
            calls.append(gr.numel())
            dist.all_reduce(gr)                        # the real RCCL call on the engine stream

        for _ in range(3):
            eng_b.train_step(allreduce=allreduce)
        eng_b.synchronize()
        assert calls == [eng_b.n_total] * 3           # exactly one collective per step over the whole flat bucket
        assert torch.equal(eng_a.flat_params, eng_b.flat_params)
        buckets = eng_b._grad_buckets                  # cut points for the overlapped variant: contiguous cover, tail first
        assert buckets[-1][1] == 0 and buckets[0][2] == eng_b.n_total
        assert all(buckets[i][1] == buckets[i + 1][2] for i in range(len(buckets) - 1))
    finally:
        dist.destroy_process_group()


def _dp_rank_main(rank, world, port, out_dir, collective, share_gpu=False):
    """one rank of test_data_parallel_two_ranks_on_two_gpus (spawned: one process per GPU; share_gpu: both on GPU 0 over gloo)"""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev_index = 0 if share_gpu else rank
    torch.set_num_threads(2)                        # (world processes x every host core otherwise)
    torch.cuda.set_device(dev_index)
    from attend_infer_repeat_amd import distributed as D
    D.init_from_env(backend="gloo" if share_gpu else "nccl")
    import gc
    import torch.distributed as dist
    keep = []
    try:
        for c in collective.split("+"):             # several protocols, one after the other, on one process group (as bench.py's A/B)
            keep.append(_dp_rank_protocol(rank, dev_index, out_dir, c))     # (engines stay alive to the end, as in a one-protocol process)
            dist.barrier()
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, "rank%d_error.txt" % rank), "w") as f:      # (a process that dies in its exit handlers loses spawn's report)
            traceback.print_exc(file=f)
        traceback.print_exc()
        raise
    dist.destroy_process_group()


def _dp_rank_protocol(rank, dev_index, out_dir, collective):
    from attend_infer_repeat_amd import distributed as D
    ocfg, B = CONFIGS["mnist_b8"]
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    fields = {f.name for f in dataclasses.fields(EngineConfig)}
    eng = AIREngine(EngineConfig(**{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields}), B,
                    device=torch.device("cuda", dev_index), seed=D.rank_seed(3, rank))
    eng.load_parameters(O.init_params(ocfg, seed=3 + rank, bias_std=0.1))     # deliberately different before the broadcast
    obs, _ = O.synthetic_batch(ocfg, B, seed=40 + rank)
    eng.set_obs(obs.cuda())
    dp = D.DataParallelEngine(eng, collective=collective)
    start = eng.flat_params.cpu().clone()
    # one step with the gradient read back before the update: split protocols expose it between the two graphs
    n_steps = int(os.environ.get("AIR_TEST_DP_STEPS") or (20 if collective == "ipc-rsag" else 3))
    for _ in range(n_steps):
        dp.train_step()
    eng.synchronize()
    in_sync = dp.replicas_in_sync()
    assert in_sync, "replicas diverged under %s" % dp.collective
    torch.save(dict(start=start, params=eng.flat_params.cpu(), grads=eng.flat_grads.cpu(), collective=dp.collective,
                    nranks=dp.rccl_nranks, noise=eng.eps_what.cpu(), steps=int(eng.step_dev.item()),
                    timed_out=bool(dp._ipc.timed_out()) if dp._ipc is not None else False,
                    ipc_local=dp._ipc.local.cpu() if dp._ipc is not None else None,
                    flags_kind=dp._ipc.flags_kind if dp._ipc is not None else None,
                    state=dp.state_dict() if os.environ.get("AIR_TEST_DP_STATE") else None),
               os.path.join(out_dir, f"{collective}_{rank}.pt"))
    dp.close()
    return eng, dp


@pytest.mark.parametrize("collective", ["torch-overlap", "torch-split", "rccl-split", "rccl-captured", "ipc-rsag"])
def test_data_parallel_two_ranks_on_two_gpus(gpu_device, tmp_path, collective):
    """The real thing where the box has it: min(device_count, 2) = 2 ranks, one process per GPU, each protocol of
    DataParallelEngine.  After three steps both replicas must hold IDENTICAL parameters (they started from rank 0's, every
    update used the same all-reduced gradient, scaled by 1/2), their summed gradient buffers must be identical, their noise
    must differ, and RCCL itself must report two ranks.  On a single-GPU box (the driver's 1-GPU tier) the torch-split protocol
    still runs, with both ranks on GPU 0 over gloo; the two RCCL protocols skip."""
    import socket
    import torch.multiprocessing as mp
    share_gpu = torch.cuda.device_count() < 2
    if share_gpu and collective.startswith("rccl-"):
        pytest.skip("the own-communicator protocols need two GPUs (RCCL refuses two ranks on one device)")
    # one GPU only: the host-issued protocol is still run for real -- two processes, two engines on GPU 0, gradients summed over
    # gloo -- so the data-parallel step (broadcast, split graphs, all-reduce in between, 1/world scaling) is exercised end to end
    port = D_free_port()
    mp.spawn(_dp_rank_main, args=(2, port, str(tmp_path), collective, share_gpu), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"{collective}_{k}.pt")) for k in range(2)]
    assert r[0]["collective"] == r[1]["collective"] == collective
    if collective == "ipc-rsag":
        # Round 5: no library collective -- the ranks map each other's buffers (hipIpc works between two
        # processes on ONE device too) and the step's graph ends with barrier | shard sum + sharded RMSProp + parameter push | barrier.
        # A two-rank sum has one order, so the parameters must be bit-equal to the plain two-graph protocol's; the replicas are
        # identical by construction; flat_grads keeps each rank's LOCAL gradient (they differ, and sum to torch-split's).
        port2 = D_free_port()
        os.environ["AIR_TEST_DP_STEPS"] = "20"                       # (the reference run takes the same twenty updates)
        try:
            mp.spawn(_dp_rank_main, args=(2, port2, str(tmp_path), "torch-split", share_gpu), nprocs=2, join=True)
        finally:
            del os.environ["AIR_TEST_DP_STEPS"]
        ref = torch.load(os.path.join(tmp_path, "torch-split_0.pt"))
        assert torch.equal(r[0]["start"], r[1]["start"]) and torch.equal(r[0]["start"], ref["start"])
        assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["params"], ref["params"])
        assert not torch.equal(r[0]["grads"], r[1]["grads"])
        assert torch.equal(r[0]["grads"] + r[1]["grads"], ref["grads"])
        assert r[0]["steps"] == r[1]["steps"] == ref["steps"] and not r[0]["timed_out"] and not r[1]["timed_out"]
        assert not torch.equal(r[0]["noise"], r[1]["noise"])
        return
    if collective == "torch-overlap":
        # the bucketed protocol (tail bucket reduced underneath the rest of the backward) must produce exactly what the plain
        # two-graph protocol produces: same launches, same sums (a two-rank sum has one order)
        port2 = D_free_port()
        mp.spawn(_dp_rank_main, args=(2, port2, str(tmp_path), "torch-split", share_gpu), nprocs=2, join=True)
        ref = torch.load(os.path.join(tmp_path, "torch-split_0.pt"))
        assert torch.equal(r[0]["params"], ref["params"]) and torch.equal(r[0]["grads"], ref["grads"])
    if not collective.startswith("torch-"):
        assert r[0]["nranks"] == r[1]["nranks"] == 2
    assert torch.equal(r[0]["start"], r[1]["start"])
    assert torch.equal(r[0]["grads"], r[1]["grads"]) and r[0]["grads"].abs().max() > 0
    assert torch.equal(r[0]["params"], r[1]["params"]) and not torch.equal(r[0]["params"], r[0]["start"])
    assert not torch.equal(r[0]["noise"], r[1]["noise"])


@pytest.fixture(scope="module")
def many_ranks_runs(tmp_path_factory):
    """world -> directory with every rank's record of the three shared-GPU protocols, from ONE spawn per world size (the processes
    run the protocols one after the other on one process group: a spawn of eight torch processes costs more than the updates)"""
    import torch.multiprocessing as mp
    done = {}

    def get(world):
        if world not in done:
            out = tmp_path_factory.mktemp("dp_world%d" % world)
            saved = {k: os.environ.get(k) for k in ("AIR_TEST_DP_STEPS", "AIR_TEST_DP_STATE")}
            os.environ.update(AIR_TEST_DP_STEPS="1", AIR_TEST_DP_STATE="1")
            try:
                mp.spawn(_dp_rank_main, args=(world, D_free_port(), str(out), "torch-split+torch-overlap+ipc-rsag", True), nprocs=world, join=True)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            done[world] = out
        return done[world]
    return get


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("collective", ["ipc-rsag", "torch-split", "torch-overlap"])
def test_data_parallel_many_ranks_sharing_one_gpu(gpu_device, many_ranks_runs, collective, world):
    """FOUR and EIGHT ranks (processes sharing GPU 0, gradients over gloo or the hipIpc mapping) -- the first
    configurations in which the order of a float sum can differ.  After ONE update from identical parameters: every replica holds
    the same bits; ipc-rsag's parameters are EXACTLY the engine's update of the host-side sum of the ranks' local gradients taken
    in rank order ((g0 + g1) + g2) + ... -- what comm_ipc.hip promises -- with every rank's slots complete after the gather
    (DataParallelEngine.state_dict); the library protocols' replicas all carry the same summed gradient and the engine's update
    of it, which agrees with the rank-order sum to rounding.  ipc-rsag's barriers (64 workgroups each) covered all eight XCDs."""
    tmp_path = many_ranks_runs(world)
    r = [torch.load(os.path.join(tmp_path, f"{collective}_{k}.pt")) for k in range(world)]
    assert all(x["collective"] == collective for x in r)
    assert all(torch.equal(x["start"], r[0]["start"]) for x in r) and all(x["steps"] == 1 for x in r)
    assert all(torch.equal(x["params"], r[0]["params"]) for x in r), "replicas differ"
    assert not torch.equal(r[0]["params"], r[0]["start"])
    # the reference update: one engine, the start parameters, a given summed gradient, grad_scale = 1 / world
    ocfg, B = CONFIGS["mnist_b8"]
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    fields = {f.name for f in dataclasses.fields(EngineConfig)}

    def update_of(gsum):
        eng = AIREngine(EngineConfig(**{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields}), B, device=torch.device("cuda", 0), seed=3)
        eng._copy_in(eng.flat_params, r[0]["start"]); eng._copy_in(eng.flat_grads, gsum)
        eng.optimizer_step(grad_scale=1.0 / world); eng.synchronize()
        return eng.flat_params.cpu(), eng.flat_mom.cpu()

    rank_order = r[0]["grads"].clone()
    if collective == "ipc-rsag":
        for k in range(1, world):
            rank_order = rank_order + r[k]["grads"]               # fp32, rank order: ((g0 + g1) + g2) + ...
        assert not any(x["timed_out"] for x in r)
        assert len({x["grads"].double().abs().sum().item() for x in r}) == world       # flat_grads keeps each rank's LOCAL gradient
        want, want_mom = update_of(rank_order)
        assert torch.equal(r[0]["params"], want)
        for x in r:                                               # complete slots on every rank after the gather
            assert torch.equal(x["state"]["flat_mom"], want_mom) and torch.equal(x["state"]["flat_params"], want)
            assert bin(int(x["ipc_local"][4])).count("1") == 8, "the barrier's workgroups did not cover the eight XCDs"
        # the flag words sit in fine-grained memory exported with the HIP IPC calls where the runtime provides it (every rank the same)
        assert len({x["flags_kind"] for x in r}) == 1
        print("ipc-rsag flag block:", r[0]["flags_kind"])
    else:
        assert all(torch.equal(x["grads"], r[0]["grads"]) for x in r)          # the all-reduced buffer, on every rank
        want, _ = update_of(r[0]["grads"])
        assert torch.equal(r[0]["params"], want)


def test_ipc_barrier_grids_cover_every_xcd(gpu_device):
    """the barrier of the ipc-rsag protocol as a grid of 16 and of 64 workgroups (one rank, its own flag block):
    every instance completes, the epochs advance, and the hardware XCC ids the workgroups recorded cover all eight XCDs -- nothing is
    assumed about which XCD workgroup i lands on; with ONE workgroup the coverage check is what it can be (no error), with the
    device's XCD count demanded of a grid that cannot give it the barrier reports err = 2."""
    import ctypes
    from attend_infer_repeat_amd import _lib, hip as H
    L, P = H.lib(), H._p
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n_wgs in (16, 64, 1):
        flags = torch.zeros(16, dtype=torch.int64, device="cuda"); local = torch.zeros(8, dtype=torch.int64, device="cuda")
        err = torch.zeros(1, dtype=torch.int64, device="cuda")
        g = torch.zeros(16, device="cuda"); pr = torch.zeros(16, device="cuda")
        peers = _lib.AirIpcPeers(); peers.world, peers.rank = 1, 0
        peers.grads[0], peers.params[0], peers.flags[0] = g.data_ptr(), pr.data_ptr(), flags.data_ptr()
        for k in range(6):
            assert L.air_dp_ipc_barrier_wgs(ctypes.byref(peers), k % 2, P(local), P(err), n_wgs, sp) == 0
        torch.cuda.synchronize()
        assert int(err.item()) == 0, (n_wgs, int(err.item()))
        assert local[0].item() == 3 and local[1].item() == 3 and local[2].item() == 6 * n_wgs and local[3].item() == 6
        xcds = bin(int(local[4].item())).count("1")
        assert xcds == (8 if n_wgs >= 16 else 1), (n_wgs, xcds)
        assert flags[0].item() == 3 and flags[8].item() == 3


def test_data_parallel_semantics_two_virtual_ranks(gpu_device):
    """the data-parallel update equals centred RMSProp on the MEAN of the per-rank gradients of independent
    per-rank reference steps.  Two ranks are emulated on one GPU: two engines with the same weights, each with its own batch
    and noise; their flat gradient buffers are summed (what the RCCL all-reduce does) and the update runs with
    grad_scale = 1/2 -- against the oracle's mean-gradient step."""
    ocfg, B = CONFIGS["mnist_b8"]
    e0, params, obs0, noise0 = make_pair(ocfg, B, seed=1, gstep=3)
    e1, _, obs1, noise1 = make_pair(ocfg, B, seed=1, gstep=3)
    obs1, _ = O.synthetic_batch(ocfg, B, seed=77)
    noise1 = O.make_noise(ocfg, B, seed=78)
    e1.set_obs(obs1.cuda()); e1.set_noise(noise1["eps_where"].cuda(), noise1["eps_what"].cuda(), noise1["u_pres"].cuda())
    for e in (e0, e1):
        e.forward(sample_noise=False); e.backward()
    e0.synchronize(); e1.synchronize()
    e0.flat_grads.add_(e1.flat_grads)                                # all_reduce(SUM) over the two ranks
    e0.optimizer_step(grad_scale=0.5)
    e0.synchronize()
    p64 = f64(params)
    _, g0 = O.forward_backward(p64, ocfg, obs0.double(), f64(noise0), global_step=3)
    _, g1 = O.forward_backward(p64, ocfg, obs1.double(), f64(noise1), global_step=3)
    mean = {k: 0.5 * (g0[k] + g1[k]) for k in g0}
    slots = O.rmsprop_init(p64)
    O.rmsprop_centered_step(p64, mean, slots, ocfg)
    for k, ref in p64.items():
        delta_ref = ref - params[k].double()
        delta = e0.params[k].cpu().double() - params[k].double()
        assert rel_err(delta, delta_ref) < 5e-3, (k, rel_err(delta, delta_ref))


def test_checkpoint_resume_is_bit_exact(gpu_device, tmp_path):
    """Save after 3 steps, resume in a fresh engine, run 3 more: identical to 6 uninterrupted steps (params, optimiser
    slots, step counter and noise stream all restored)."""
    ocfg, B = CONFIGS["mnist_b8"]
    eng_a, *_ = make_pair(ocfg, B, seed=5, gstep=0)
    eng_a.capture()
    for _ in range(3):
        eng_a.train_step()
    path = str(tmp_path / "ckpt.pt")
    torch.save(eng_a.state_dict(), path)
    for _ in range(3):
        eng_a.train_step()
    eng_b, *_ = make_pair(ocfg, B, seed=99, gstep=0)            # different init, then restored
    eng_b.load_state_dict(torch.load(path))
    eng_b.set_obs(eng_a.obs.clone())                            # the data batch is not part of a checkpoint
    eng_b.capture()
    for _ in range(3):
        eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    assert eng_b.global_step == eng_a.global_step == 6
    assert torch.equal(eng_a.flat_params, eng_b.flat_params) and torch.equal(eng_a.flat_mom, eng_b.flat_mom)


@pytest.mark.parametrize("name,kw,B,mfma,launches", [
    ("configs[1]", {}, 64, "f32", 29),
    ("configs[3]", dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 64, "f32", 35),
    ("configs[4]", {}, 1024, "bf16", 39),
])
def test_launches_per_train_step_of_the_named_configurations(gpu_device, name, kw, B, mfma, launches):
    """The dependent-launch count of the single-GPU train step is what the latency regime is optimised for (DESIGN section 3, round 5:
    34 -> 29 at configs[1]); a plan change that adds a launch should be a decision, not an accident.  The entry of the BPTT rides in
    its first link in the latency regime (air_lstm_step_bwd_entry), not in the throughput regime."""
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    eng = AIREngine(EngineConfig(mfma_dtype=mfma, **kw), B, seed=1, keep_canvas_steps=True)
    assert sum(eng.kernel_launch_count().values()) == launches, (name, eng.kernel_launch_count())
    names = [n for plan in eng._single_gpu_step_plans() for _, _, n in plan]
    assert ("air_lstm_step_bwd_entry" in names) == (B == 64), name
    assert ("air_lstm_pointwise_bwd" in names or "air_lstm_pointwise_bwd_opt" in names or "air_lstm_pointwise_bwd_bf16" in names) == (B != 64), name


@pytest.mark.gpu
def test_tf_checkpoint_export_import_between_engines(gpu_device, tmp_path):
    """an engine's parameters written as a TF-1 `model.ckpt` (TensorBundle, tf_checkpoint.write_bundle) under
    reference-like variable names -- with the RMSProp slots and global_step a Saver writes beside them -- and imported into a differently
    initialised engine: same parameters and optimiser state bit for bit, and the next train step (same batch, same noise state) leaves
    both on identical parameters."""
    from attend_infer_repeat_amd import tf_checkpoint as TF
    from test_tf_checkpoint import _reference_like_checkpoint
    ocfg, B = CONFIGS["mnist_b8"]
    eng_a, *_ = make_pair(ocfg, B, seed=5, gstep=0)
    eng_b, *_ = make_pair(ocfg, B, seed=99, gstep=0)
    eng_a.synchronize(); eng_b.synchronize()
    assert not torch.equal(eng_a.flat_params, eng_b.flat_params)
    names, _ = _reference_like_checkpoint(eng_a.cfg, np.random.default_rng(0))       # only the NAMES of this helper are used
    tfmap = TF.default_name_map(eng_a.param_shapes, {k: v.shape for k, v in names.items()})
    assert set(tfmap) == set(eng_a.param_shapes)
    eng_a.train_step(); eng_a.synchronize()                                           # (so that the slots are not their initial values)
    tensors = {tfmap[k]: eng_a.params[k].detach().cpu().numpy() for k in eng_a.param_shapes}
    for flat, suffix in ((eng_a.flat_ms, "RMSProp"), (eng_a.flat_mg, "RMSProp_1"), (eng_a.flat_mom, "RMSProp_2")):
        for k in eng_a.param_shapes:
            off, n = eng_a.param_offsets[k], eng_a.param_sizes[k]
            tensors[tfmap[k] + "/" + suffix] = flat[off:off + n].cpu().numpy().reshape(eng_a.param_shapes[k])
    tensors["global_step"] = np.int64(1)
    prefix = str(tmp_path / "model.ckpt-1")
    TF.write_bundle(prefix, tensors)
    named = TF.import_tf_checkpoint(prefix, eng_b.param_shapes)
    eng_b.load_parameters({k: torch.from_numpy(v) for k, v in named.items()})
    eng_b.reset_optimizer()
    slots = TF.import_tf_optimizer_slots(prefix, eng_b.param_shapes)
    eng_b.load_optimizer_slots(**{s: {k: torch.from_numpy(v) for k, v in d.items()} for s, d in slots.items()})
    eng_b.set_global_step(TF.global_step_of(prefix))
    eng_b.set_obs(eng_a.obs.clone())
    eng_a.synchronize(); eng_b.synchronize()
    eng_b.rng_state.copy_(eng_a.rng_state)
    torch.cuda.synchronize()
    assert torch.equal(eng_a.flat_params, eng_b.flat_params)
    for k in eng_a.param_shapes:                                                      # (the 16-byte padding between tensors is not a slot)
        off, n = eng_a.param_offsets[k], eng_a.param_sizes[k]
        assert all(torch.equal(fa[off:off + n], fb[off:off + n]) for fa, fb in
                   ((eng_a.flat_ms, eng_b.flat_ms), (eng_a.flat_mg, eng_b.flat_mg), (eng_a.flat_mom, eng_b.flat_mom))), k
    assert eng_b.global_step == eng_a.global_step == 1
    eng_a.train_step(); eng_b.train_step()
    eng_a.synchronize(); eng_b.synchronize()
    assert torch.equal(eng_a.flat_params, eng_b.flat_params)
