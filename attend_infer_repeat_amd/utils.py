"""Small host-side helpers."""


class AttrDict(dict):
    """dict with attribute access -- stands in for attrdict.AttrDict, which the reference's training script uses for
    the prior specifications (scripts/multi_mnist.py:40-51: `'loc' in where_shift_prior` and `where_shift_prior.loc`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
