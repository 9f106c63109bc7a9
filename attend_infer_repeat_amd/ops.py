"""Loss bookkeeping helpers with the interface of the reference's ops.py (attend_infer_repeat/ops.py:5-76)."""
import torch


class Loss(object):
    """Accumulates a scalar `value` and a `per_sample` vector with weights (ops.py:5-43)."""

    def __init__(self):
        self._value = None
        self._per_sample = None

    def add(self, loss=None, per_sample=None, weight=1.):
        if isinstance(loss, Loss):
            per_sample = loss.per_sample
            loss = loss.value
        self._update('_value', loss, weight)
        self._update('_per_sample', per_sample, weight)

    def _update(self, name, expr, weight):
        if expr is None:
            return
        value = getattr(self, name)
        expr = expr * weight
        if value is None:
            value = expr
        else:
            assert tuple(value.shape) == tuple(expr.shape), \
                'Shape should be {} but is {}'.format(tuple(value.shape), tuple(expr.shape))
            value = value + expr
        setattr(self, name, value)

    def _get_value(self, name):
        v = getattr(self, name)
        return torch.zeros([]) if v is None else v

    @property
    def value(self):
        return self._get_value('_value')

    @property
    def per_sample(self):
        return self._get_value('_per_sample')


class MovingAverage(object):
    """State of make_moving_average: a non-trainable variable updated as var <- decay*var + (1-decay)*value."""

    def __init__(self, name, init, decay):
        self.name, self.decay = name, float(decay)
        self.var = None
        self._init = float(init)

    def update(self, value):
        value = value.detach()
        if self.var is None:
            self.var = torch.full_like(value, self._init)
        self.var = self.decay * self.var + (1.0 - self.decay) * value
        return self.var


def make_moving_average(name, value, init, decay, log=True, store=None, update=True):
    """Exp-moving average of `value` (ops.py:46-64).  In the reference the update is an UPDATE_OP that runs only with the
    train step and the returned tensor is the *variable* (pre-update value on the first step = init).
    `store`: the dict that owns the state (a tf.Graph's variable collection in the reference; here the model instance's
    `_moving_averages`, so two models never share statistics).  `update=False` reads the variable without running the
    update op (evaluation passes)."""
    if store is None:
        store = {}
    ma = store.get(name)
    if ma is None:
        ma = store[name] = MovingAverage(name, init, decay)
    prev = ma.var if ma.var is not None else torch.full_like(value.detach(), float(init))
    if update:
        ma.update(value)
    return prev


def clip_preserve(expr, min, max):
    """Clips the value but preserves the chain rule (ops.py:67-76)."""
    # tf.clip_by_value order: min with the upper bound first, then max with the lower bound (the lower bound wins,
    # which is what makes `clip_preserve(prob, 1e-32, prob)` at prior.py:150 a pure lower clip)
    as_t = lambda v: v if torch.is_tensor(v) else torch.as_tensor(v, dtype=expr.dtype, device=expr.device)
    clipped = torch.maximum(torch.minimum(expr, as_t(max)), as_t(min))
    return (clipped - expr).detach() + expr
