"""Logging / evaluation helpers with the interface of the reference's evaluation.py (make_logger, make_expr_logger,
rect_stn, make_fig; attend_infer_repeat/evaluation.py:14-166).  Pure consumers of the model's attributes: no kernels
here.  TensorBoard summaries are replaced by returned dicts (and an optional JSON-lines file); figures need matplotlib.
"""
import json
import time


def attention_box(stn_params, width, height):
    """Pixel rectangle (left, top, box_width, box_height) that the glimpse of [sx, tx, sy, ty] covers on a width x height
    canvas.  This is the only in-tree statement of the reference's spatial-transformer convention (evaluation.py:23-28): the
    glimpse spans sx (sy) of the canvas extent and its centre sits tx (ty) half-extents away from the canvas centre."""
    sx, tx, sy, ty = (float(v) for v in stn_params)
    return width * (1. - sx + tx) / 2, height * (1. - sy + ty) / 2, width * sx, height * sy


def rect_stn(ax, width, height, stn_params, c=None, line_width=3):
    """Draw the attention box of one glimpse on `ax` (pixel centres at integer coordinates, hence the half-pixel shift)."""
    from matplotlib.patches import Rectangle
    left, top, bw, bh = attention_box(stn_params, width, height)
    patch = Rectangle((left - .5, top - .5), bw, bh, linewidth=line_width, edgecolor=c, facecolor='none')
    ax.add_patch(patch)
    return patch


def _figure_tensors(air):
    """what the progress figure shows, as host arrays"""
    host = lambda t: t.detach().cpu().numpy()
    step_probs = None
    if hasattr(air, "num_steps_distrib"):
        step_probs = host(air.num_steps_distrib.prob()[..., 1:])          # [B, T]: p(n = t + 1)
    return dict(obs=host(air.obs), canvas=host(air.canvas), glimpse=host(air.glimpse), presence=host(air.presence)[..., 0],
                where=host(air.where), step_probs=step_probs)


def make_fig(air, checkpoint_dir=None, global_step=None, n_samples=10):
    """Progress figure with the content of the reference's (evaluation.py:31-65): one column per sample; the first row shows the
    input, the next max_steps rows the canvas after each step with the attention box of every step that is present, the last
    max_steps rows the glimpse each step reconstructed, titled with the sampled presence and the posterior probability of that
    step count.  Saved as progress_fig_<global_step>.png when a directory is given."""
    import os.path as osp
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    d = _figure_tensors(air)
    T = air.max_steps
    cols = min(n_samples, air.batch_size)
    img_h, img_w = d["obs"].shape[1:]
    inch = 1.5
    fig, axes = plt.subplots(2 * T + 1, cols, figsize=(inch * cols, inch * (2 * T + 1)), squeeze=False)
    for col in range(cols):
        axes[0][col].imshow(d["obs"][col], cmap='gray', vmin=0, vmax=1)
        for t in range(T):
            canvas_ax, glimpse_ax = axes[1 + t][col], axes[1 + T + t][col]
            canvas_ax.imshow(d["canvas"][t, col], cmap='gray', vmin=0, vmax=1)
            if d["presence"][t, col] > .5:
                rect_stn(canvas_ax, img_w, img_h, d["where"][t, col], 'r')
            glimpse_ax.imshow(d["glimpse"][t, col], cmap='gray')
            if d["step_probs"] is not None:
                glimpse_ax.set_title('{:d} with p({:d}) = {:.02f}'.format(int(d["presence"][t, col]), t + 1,
                                                                         float(d["step_probs"][col, t])), fontsize=4 * inch)
    for ax in axes.ravel():
        ax.set_axis_off()
    if checkpoint_dir is not None:
        fig.savefig(osp.join(checkpoint_dir, 'progress_fig_{}.png'.format(global_step)), dpi=300)
        plt.close(fig)
    return fig


def gradient_summaries(named_grads, named_vars, norm=True, ratio=True, histogram=False, bins=30):
    """evaluation.py:221-248: the global norm of the gradient, per variable mean(|g| / (|v| + 1e-8)) (log_ratio,
    evaluation.py:169-180) and -- histogram=True, the reference's default -- a histogram of every gradient tensor, the content of its
    `tf.summary.histogram('grad_hist/<name>', g)` as plain data: {'counts': [...], 'edges': [...]} over `bins` equal-width bins between
    the tensor's min and max (TensorBoard's own compression of the same values is a display artefact).  `named_grads` / `named_vars`:
    name -> tensor (AIREngine.named_grads() / .params; on the engine the gradients are those of the last update and the variables the
    ones it produced).  Returns {'grad_norm': float, 'grad_ratio/<name>': float, 'grad_hist/<name>': dict}.  Off by default here
    because the script logs JSON lines, not event files: scripts/multi_mnist.py --grad-histograms switches it on."""
    import torch
    out = {}
    if norm:
        out['grad_norm'] = float(torch.sqrt(sum((g.double() ** 2).sum() for g in named_grads.values())))
    for k, g in named_grads.items():
        if ratio:
            out['grad_ratio/' + k] = float((g.abs() / (named_vars[k].abs() + 1e-8)).mean())
        if histogram:
            gf = g.detach().float().reshape(-1)
            fin = gf[torch.isfinite(gf)]
            lo, hi = (float(fin.min()), float(fin.max())) if fin.numel() else (0.0, 0.0)
            if hi <= lo:
                hi = lo + 1e-12
            counts = torch.histc(fin, bins=bins, min=lo, max=hi) if fin.numel() else torch.zeros(bins)
            out['grad_hist/' + k] = {'counts': [int(c) for c in counts.tolist()],
                                     'edges': [lo + (hi - lo) * i / bins for i in range(bins + 1)],
                                     'non_finite': int(gf.numel() - fin.numel())}
    return out


def step_summaries(air, histogram=False):
    """The scalars the reference registers with tf.summary.scalar and writes every 1000 iterations (multi_mnist.py:138-140;
    model.py:152,185,213,244-257,323-371), read from the engine after a train step: the objective's terms on the batch just
    trained on + the gradient summaries of that update."""
    eng = air._engine
    o = eng.outputs()
    pick = ["rec_loss", "kl_num_steps", "kl_what", "kl_where", "prior_loss", "loss", "opt_loss"]
    if eng.cfg.use_reinforce:
        pick += ["imp_weight_mean", "imp_weight_var", "reinforce_loss", "baseline_loss"]
    out = {('rec' if k == 'rec_loss' else 'prior' if k == 'prior_loss' else k): _scalar(o[k]) for k in pick}
    out['num_step'] = _scalar(o["num_step_per_sample"].mean())
    out.update(gradient_summaries(eng.named_grads(), eng.params, histogram=histogram))
    return out


def _scalar(v):
    return float(v.item()) if hasattr(v, "item") else float(v)


def logged_exprs(air):
    """The scalar set of evaluation.py:69-92, as name -> callable(air) (eager mode: attributes are refreshed per pass)."""
    exprs = {
        'loss': lambda a: a.loss.value,
        'rec_loss': lambda a: a.rec_loss,
        'num_step_acc': lambda a: a.num_step_accuracy,
        'num_step': lambda a: a.num_step,
    }
    if air.use_prior:
        exprs['prior_loss'] = lambda a: a.prior_loss.value
        if air.num_steps_prior is not None:
            exprs['kl_num_steps'] = lambda a: a.kl_num_steps
        if air.what_prior is not None:
            exprs['kl_what'] = lambda a: a.kl_what
            exprs['kl_where'] = lambda a: a.kl_where
    if air.use_reinforce:
        if air.baseline is not None:
            exprs['baseline_loss'] = lambda a: a.baseline_loss
        exprs['reinforce_loss'] = lambda a: a.reinforce_loss
        exprs['imp_weight'] = lambda a: a.importance_weight.mean()
    return exprs


def make_expr_logger(air, data_fn, num_batches, expr_dict, name, writer=None, measure_time=True):
    """evaluation.py:112-166: average `expr_dict` over `num_batches` evaluation passes on batches from `data_fn`."""
    def logger(itr=0, num_batches_to_eval=None, write=True):
        n = num_batches if num_batches_to_eval is None else num_batches_to_eval
        n = max(int(n), 1)
        acc = {k: 0. for k in expr_dict}
        start = time.time()
        for _ in range(n):
            obs, nums = data_fn()
            air.evaluate(obs, nums)
            for k, fn in expr_dict.items():
                acc[k] += _scalar(fn(air))
        acc = {k: v / n for k, v in acc.items()}
        t = time.time() - start
        msg = 'Step {}, Data {} '.format(itr, name) + ', '.join('{} = {:.4f}'.format(k, v) for k, v in acc.items())
        if measure_time:
            msg += ', eval time = {:.4}s'.format(t)
        print(msg)
        if write and writer is not None:
            writer.write(json.dumps(dict(step=int(itr), data=name, **acc)) + "\n"); writer.flush()
        return acc
    return logger


def make_logger(air, train_data_fn, train_batches, test_data_fn, test_batches, writer=None):
    """evaluation.py:68-109.  (The reference divides the batch counts by batch_size a second time, evaluation.py:94,100
    -- SURVEY B-8; here `*_batches` is simply the number of batches to average over.)"""
    exprs = logged_exprs(air)
    train_log = make_expr_logger(air, train_data_fn, train_batches, exprs, 'train', writer)
    test_log = make_expr_logger(air, test_data_fn, test_batches, exprs, 'test', writer)

    def log(train_itr):
        a = train_log(train_itr)
        b = test_log(train_itr)
        print()
        return a, b
    return log
