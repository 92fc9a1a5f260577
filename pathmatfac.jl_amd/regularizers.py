"""Regularizers (src/regularizers.jl, src/featureset_ard.jl): host-side descriptors.

Their value and gradient are evaluated on the device by k_reg_step (csrc/pmf_hip.hip); the classes below hold the
parameters, mirror the reference's constructors and freeze bookkeeping, and know how to marshal themselves
through the C ABI.  Network / L1 / SelectiveL1 regularizers are out of scope and raise."""
import numpy as np

from .layers import BatchScale, BatchShift, ColScale, ColShift, FrozenLayer, Identity
from .util import UnitRange, featuresets_to_dense, ids_to_ranges, unique


class ZeroReg:
    """`x -> 0` (regularizers.jl:670, fit.jl:769, transform.jl:61,70)."""

    def add_to(self, ctx, which, p=1.0):
        pass


class L2Regularizer:  # regularizers.jl:11-55
    def __init__(self, K_or_weights, w=None):
        self.weights = (np.full(int(K_or_weights), float(w), dtype=np.float32) if w is not None
                        else np.asarray(K_or_weights, dtype=np.float32))

    def add_to(self, ctx, which, p=1.0):
        ctx.add_reg_l2(which, self.weights, p)


class GroupRegularizer:  # regularizers.jl:345-359
    def __init__(self, group_labels=None, weight=1.0, K=1, group_idx=None, group_weights=None, labels=None):
        if group_idx is not None:
            self.group_labels = list(labels) if labels is not None else list(range(len(group_idx)))
            self.group_idx = tuple(UnitRange(*g) for g in group_idx)
            self.group_weights = tuple(np.asarray(w, dtype=np.float32) for w in group_weights)
        else:
            self.group_labels = unique(group_labels)
            self.group_idx = tuple(ids_to_ranges(group_labels))
            self.group_weights = tuple(np.full(K, weight, dtype=np.float32) for _ in self.group_labels)

    def add_to(self, ctx, which, p=1.0):
        ctx.add_reg_group(which, [(g.start, g.stop) for g in self.group_idx], np.stack(self.group_weights), p)


class ARDRegularizer:  # regularizers.jl:526-543
    def __init__(self, column_groups, alpha=np.float32(1.001), beta=np.float32(0.001), weight=1.0):
        self.col_ranges = tuple(ids_to_ranges(column_groups))
        self.alpha = tuple(np.float32(alpha) for _ in self.col_ranges)
        self.beta = tuple(np.float32(beta) for _ in self.col_ranges)
        self.weight = weight

    def reweight_eb_(self, X=None):  # regularizers.jl:588-609 (Q12: hard-sets alpha = beta = 0.001)
        self.alpha = tuple(np.float32(0.001) for _ in self.col_ranges)
        self.beta = tuple(np.float32(0.001) for _ in self.col_ranges)

    def add_to(self, ctx, which, p=1.0):
        if which != "Y":
            raise NotImplementedError("ARDRegularizer is supported on Y only")
        ctx.add_yreg_ard([(r.start, r.stop) for r in self.col_ranges], self.alpha, self.beta, p)


class FeatureSetARDReg:  # featureset_ard.jl:19-65
    def __init__(self, K, feature_views, S_vec, featureset_ids_vec, alpha0=1.01, v0=0.8, lr=0.05):
        N = len(feature_views)
        self.col_ranges = tuple(ids_to_ranges(feature_views))
        for cr, S, fid in zip(self.col_ranges, S_vec, featureset_ids_vec):
            assert len(cr) == S.shape[1], "The number of columns in each `feature_view` must match size(S, 2)"
            assert len(fid) == S.shape[0], "Number of featureset_ids incompatible with size(S, 1)"
        self.S = tuple(np.asarray(S, dtype=np.float32) for S in S_vec)
        self.A = tuple(np.zeros((S.shape[0], K), dtype=np.float32) for S in self.S)
        self.lambda_ = tuple(np.ones(K, dtype=np.float32) for _ in self.S)
        self.alpha0 = np.float32(alpha0)
        self.v0 = np.float32(v0)
        self.lr = np.float32(lr)
        self.featureset_ids = tuple(list(f) for f in featureset_ids_vec)
        self.alpha = np.full(N, self.alpha0, dtype=np.float32)               # :53
        self.beta = np.full((K, N), self.alpha0 - np.float32(1), dtype=np.float32)  # :54

    def add_to(self, ctx, which, p=1.0):
        if which != "Y":
            raise NotImplementedError("FeatureSetARDReg is supported on Y only")
        ctx.add_yreg_fsard(self.alpha, self.beta, p)


def construct_featureset_ard(K, feature_ids, feature_views, feature_sets_dict, featureset_ids=None,
                             alpha0=np.float32(1.001), v0=np.float32(0.8), lr=np.float32(0.05)):
    """featureset_ard.jl:112-132.  `feature_sets_dict` maps view -> list of feature sets (a list indexed by view
    position is accepted too, as in test/runtests.jl:826)."""
    col_ranges = ids_to_ranges(feature_views)
    unq_views = unique(feature_views)

    def sets_of(i, uv):
        return feature_sets_dict[uv] if isinstance(feature_sets_dict, dict) else feature_sets_dict[i]
    if featureset_ids is None:
        featureset_ids = {uv: list(range(1, len(sets_of(i, uv)) + 1)) for i, uv in enumerate(unq_views)}
    S_vec, fs_vec = [], []
    for i, (cr, uv) in enumerate(zip(col_ranges, unq_views)):
        ids = list(feature_ids[cr.start - 1:cr.stop])
        S_vec.append(featuresets_to_dense(ids, sets_of(i, uv)))
        fs_vec.append(featureset_ids[uv])
    return FeatureSetARDReg(K, feature_views, S_vec, fs_vec, alpha0=alpha0, v0=v0, lr=lr)


class CompositeRegularizer:  # regularizers.jl:616-643
    def __init__(self, regularizers, mixture_p):
        self.regularizers = tuple(regularizers)
        self.mixture_p = tuple(float(p) for p in mixture_p)

    def add_to(self, ctx, which, p=1.0):
        for r, q in zip(self.regularizers, self.mixture_p):
            r.add_to(ctx, which, p * q)


def construct_composite_reg(regs, mixture_p):  # regularizers.jl:625-631
    if len(regs) == 0:
        regs, mixture_p = [ZeroReg()], [0]
    return CompositeRegularizer(regs, mixture_p)


def construct_X_reg(K, M, sample_ids, sample_conditions, sample_graphs, lambda_X_l2, lambda_X_condition,
                    lambda_X_graph, Y_ard, Y_geneset_ard):
    """regularizers.jl:655-689."""
    if Y_ard or Y_geneset_ard:
        if sample_conditions is not None:
            return GroupRegularizer(sample_conditions, weight=1.0, K=K)
        return L2Regularizer(K, 1.0)
    regs, mix = [ZeroReg(), ZeroReg(), ZeroReg()], np.zeros(3)
    if lambda_X_l2 is not None:
        regs[0], mix[0] = L2Regularizer(K, lambda_X_l2), 1
    if sample_conditions is not None:
        regs[1], mix[1] = GroupRegularizer(sample_conditions, weight=lambda_X_condition, K=K), 1
    if sample_graphs is not None:
        raise NotImplementedError("NetworkRegularizer (sample_graphs) is out of scope of the HIP path (SURVEY section 2)")
    mix = mix / mix.sum() if mix.sum() > 0 else mix      # the reference divides by zero here (NaN weights on x->0 terms)
    return construct_composite_reg(regs, mix)


def construct_Y_reg(K, N, feature_ids, feature_views, feature_sets_dict, feature_graphs, lambda_Y_l2,
                    lambda_Y_selective_l1, lambda_Y_graph, Y_ard, Y_geneset_ard, featureset_names, alpha0, v0):
    """regularizers.jl:696-739."""
    if Y_geneset_ard:
        return construct_featureset_ard(K, feature_ids, feature_views, feature_sets_dict,
                                        featureset_ids=featureset_names, alpha0=alpha0, v0=v0)
    if Y_ard:
        return ARDRegularizer(feature_views)
    regs, mix = [ZeroReg(), ZeroReg(), ZeroReg()], np.zeros(3)
    if lambda_Y_l2 is not None:
        regs[0], mix[0] = GroupRegularizer(feature_views, K=K, weight=lambda_Y_l2), 1
    if feature_ids is not None and feature_graphs is not None:
        if lambda_Y_selective_l1 is not None or lambda_Y_graph is not None:
            raise NotImplementedError("SelectiveL1Reg / NetworkRegularizer (feature_graphs) are out of scope of the HIP path")
    s = mix.sum() or 1
    return construct_composite_reg(regs, mix / s)


class ColParamReg:  # regularizers.jl:462-479
    def __init__(self, feature_views, weight=1.0, center=0.0):
        self.col_ranges = tuple(ids_to_ranges(feature_views))
        self.weights = tuple(float(weight) for _ in self.col_ranges)
        self.centers = tuple(float(center) for _ in self.col_ranges)


class BatchArrayReg:  # regularizers.jl:781-792
    def __init__(self, ba, center=0.0, weight=1.0):
        nbs = [v.shape[0] for v in ba.values]
        self.centers = tuple(np.full(n, center, dtype=np.float32) for n in nbs)
        self.weights = tuple(np.full(n, weight, dtype=np.float32) for n in nbs)


class FrozenRegularizer:  # regularizers.jl:950-969
    def __init__(self, reg):
        self.reg = reg


class SequenceReg:  # regularizers.jl:896-905
    def __init__(self, regs):
        self.regs = tuple(regs)

    def set_reg_(self, idx, reg):
        rs = list(self.regs)
        rs[idx - 1] = reg
        self.regs = tuple(rs)

    def frozen_mask(self):
        return sum(1 << i for i, r in enumerate(self.regs) if isinstance(r, FrozenRegularizer))


def construct_layer_reg(feature_views, batch_dict, layers, lambda_layer):
    """regularizers.jl:908-926."""
    regs = [ZeroReg(), ZeroReg(), ZeroReg(), ZeroReg()]
    if feature_views is not None:
        regs[0] = ColParamReg(feature_views, weight=lambda_layer)
        regs[2] = ColParamReg(feature_views, weight=lambda_layer)
    if batch_dict is not None:
        regs[1] = BatchArrayReg(layers.layers[1].logdelta, weight=lambda_layer)
        regs[3] = BatchArrayReg(layers.layers[3].theta, weight=lambda_layer)
    return SequenceReg(regs)


def _idx_list(idx):
    return [idx] if np.isscalar(idx) else list(idx)


def freeze_reg_(sr, idx):  # freeze_reg! regularizers.jl:972-985
    for i in _idx_list(idx):
        if not isinstance(sr.regs[i - 1], FrozenRegularizer):
            sr.set_reg_(i, FrozenRegularizer(sr.regs[i - 1]))


def unfreeze_reg_(sr, idx):  # unfreeze_reg! regularizers.jl:987-999
    for i in _idx_list(idx):
        if isinstance(sr.regs[i - 1], FrozenRegularizer):
            sr.set_reg_(i, sr.regs[i - 1].reg)
