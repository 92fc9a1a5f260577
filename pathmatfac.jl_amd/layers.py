"""Column layers (src/layers.jl): host-side structs only.  Their forward and pull-back are fused into the HIP
data pass; these classes carry the parameters to and from the device and mirror the reference's
freeze / view bookkeeping."""
import numpy as np

from .batch_array import BatchArray
from .util import ids_to_ranges, unique


class Identity:
    """`x -> x` (layers.jl:243; transform.jl:64-65)."""

    def view(self, idx1, idx2):
        return self


class ColScale:  # layers.jl:9-31
    def __init__(self, N_or_logsigma):
        self.logsigma = (np.zeros(int(N_or_logsigma)) if np.isscalar(N_or_logsigma)
                         else np.asarray(N_or_logsigma, dtype=np.float64))

    def view(self, idx1, idx2):
        return ColScale(self.logsigma[idx2[0] - 1:idx2[1]])


class ColShift:  # layers.jl:53-75
    def __init__(self, N_or_mu, rng=None):
        if np.isscalar(N_or_mu):
            rng = rng or np.random.default_rng()
            self.mu = rng.standard_normal(int(N_or_mu)) * 1e-4      # layers.jl:60
        else:
            self.mu = np.asarray(N_or_mu, dtype=np.float64)

    def view(self, idx1, idx2):
        return ColShift(self.mu[idx2[0] - 1:idx2[1]])


def _zero_batch_array(col_batches, batch_dict):
    """layers.jl:101-116 / 164-179: zero-valued BatchArray over the views that have row batches."""
    unq = unique(col_batches)
    ranges = ids_to_ranges(col_batches)
    values = [dict() for _ in unq]
    for k, (cbi, cr) in enumerate(zip(unq, ranges)):
        if cbi in batch_dict:
            for rb in unique(batch_dict[cbi]):
                values[k][rb] = np.zeros(len(cr))
    return BatchArray.from_views(col_batches, batch_dict, values)


class BatchScale:  # layers.jl:95-152
    def __init__(self, col_batches=None, batch_dict=None, logdelta=None):
        self.logdelta = logdelta if logdelta is not None else _zero_batch_array(col_batches, batch_dict)

    def view(self, idx1, idx2):
        if idx2 is None:
            stp = self.logdelta.col_ranges[-1].stop if self.logdelta.col_ranges else 0
            idx2 = (1, stp)
        return BatchScale(logdelta=self.logdelta.view(idx1, idx2))


class BatchShift:  # layers.jl:158-214
    def __init__(self, col_batches=None, batch_dict=None, theta=None):
        self.theta = theta if theta is not None else _zero_batch_array(col_batches, batch_dict)

    def view(self, idx1, idx2):
        if idx2 is None:
            stp = self.theta.col_ranges[-1].stop if self.theta.col_ranges else 0
            idx2 = (1, stp)
        return BatchShift(theta=self.theta.view(idx1, idx2))


class FrozenLayer:  # layers.jl:299-334
    def __init__(self, layer):
        self.layer = layer

    def view(self, idx1, idx2):
        return FrozenLayer(self.layer.view(idx1, idx2))


class ViewableComposition:  # layers.jl:221-259
    def __init__(self, layers):
        self.layers = tuple(layers)

    def view(self, idx1, idx2):
        return ViewableComposition([l.view(idx1, idx2) for l in self.layers])

    def set_layer_(self, idx, layer):  # set_layer! layers.jl:255-259 (1-based idx)
        ls = list(self.layers)
        ls[idx - 1] = layer
        self.layers = tuple(ls)

    def unwrapped(self, idx):
        l = self.layers[idx - 1]
        return l.layer if isinstance(l, FrozenLayer) else l

    def frozen_mask(self):
        """bit (l-1) set <=> layer l is a FrozenLayer (what pmf_fit_opts.frozen_layers takes)."""
        return sum(1 << i for i, l in enumerate(self.layers) if isinstance(l, FrozenLayer))


def construct_model_layers(feature_views, batch_dict, rng=None):
    """layers.jl:240-253: [ColScale, BatchScale | x->x, ColShift, BatchShift | x->x]."""
    N = len(feature_views)
    layers = [ColScale(N), Identity(), ColShift(N, rng=rng), Identity()]
    if batch_dict is not None:
        layers[1] = BatchScale(feature_views, batch_dict)
        layers[3] = BatchShift(feature_views, batch_dict)
    return ViewableComposition(layers)


def _idx_list(idx):
    return [idx] if np.isscalar(idx) else list(idx)


def freeze_layer_(vc, idx):  # freeze_layer! layers.jl:337-349 ; Functions are never wrapped (:310-312)
    for i in _idx_list(idx):
        l = vc.layers[i - 1]
        if not isinstance(l, (FrozenLayer, Identity)):
            vc.set_layer_(i, FrozenLayer(l))


def unfreeze_layer_(vc, idx):  # unfreeze_layer! layers.jl:351-363
    for i in _idx_list(idx):
        l = vc.layers[i - 1]
        if isinstance(l, FrozenLayer):
            vc.set_layer_(i, l.layer)
