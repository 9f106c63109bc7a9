"""Logging / evaluation helpers with the interface of the reference's evaluation.py (make_logger, make_expr_logger,
rect_stn, make_fig; attend_infer_repeat/evaluation.py:14-166).  Pure consumers of the model's attributes: no kernels
here.  TensorBoard summaries are replaced by returned dicts (and an optional JSON-lines file); figures need matplotlib.
"""
import json
import time


def rect_stn(ax, width, height, stn_params, c=None, line_width=3):
    """Draw the attention box of [sx, tx, sy, ty] (evaluation.py:23-28: the in-tree statement of the ST convention)."""
    from matplotlib.patches import Rectangle
    sx, tx, sy, ty = (float(v) for v in stn_params)
    x = width * (1. - sx + tx) / 2
    y = height * (1. - sy + ty) / 2
    r = Rectangle((x - .5, y - .5), width * sx, height * sy, linewidth=line_width, edgecolor=c, facecolor='none')
    ax.add_patch(r)
    return r


def make_fig(air, checkpoint_dir=None, global_step=None, n_samples=10):
    """Progress figure (evaluation.py:31-65): inputs, per-step canvases with attention boxes, per-step glimpses."""
    import os.path as osp
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import numpy as np
    n_steps = air.max_steps
    xx = air.obs.detach().cpu().numpy()
    pred_canvas = air.canvas.detach().cpu().numpy()
    pred_crop = air.glimpse.detach().cpu().numpy()
    prob = air.num_steps_distrib.prob()[..., 1:].detach().cpu().numpy() if hasattr(air, "num_steps_distrib") else None
    pres = air.presence.detach().cpu().numpy()
    w = air.where.detach().cpu().numpy()
    height, width = xx.shape[1:]
    bs = min(n_samples, air.batch_size)
    scale = 1.5
    fig, axes = plt.subplots(2 * n_steps + 1, bs, figsize=scale * np.asarray((bs, 2 * n_steps + 1)))
    for i, ax in enumerate(axes[0]):
        ax.imshow(xx[i], cmap='gray', vmin=0, vmax=1)
    for i, ax_row in enumerate(axes[1:1 + n_steps]):
        for j, ax in enumerate(ax_row):
            ax.imshow(pred_canvas[i, j], cmap='gray', vmin=0, vmax=1)
            if pres[i, j, 0] > .5:
                rect_stn(ax, width, height, w[i, j], 'r')
    for i, ax_row in enumerate(axes[1 + n_steps:]):
        for j, ax in enumerate(ax_row):
            ax.imshow(pred_crop[i, j], cmap='gray')
            if prob is not None:
                ax.set_title('{:d} with p({:d}) = {:.02f}'.format(int(pres[i, j, 0]), i + 1, float(prob[j, i])),
                             fontsize=4 * scale)
    for ax in axes.flatten():
        ax.xaxis.set_visible(False); ax.yaxis.set_visible(False)
    if checkpoint_dir is not None:
        fig.savefig(osp.join(checkpoint_dir, 'progress_fig_{}.png'.format(global_step)), dpi=300)
        plt.close('all')
    return fig


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
