"""Index / ID bookkeeping.  Bit-exact restatement of the reference's integer utilities, pinned by
the reference's own known answers (tests/golden/util_index.json <- test/runtests.jl:19-56).

Ranges are the reference's Julia `UnitRange`s: 1-based, inclusive on both ends."""
from collections import namedtuple

import numpy as np


class UnitRange(namedtuple("UnitRange", ["start", "stop"])):
    """Julia `start:stop` (1-based inclusive).  An empty range has stop == start - 1."""
    __slots__ = ()

    def __len__(self):
        return max(0, self.stop - self.start + 1)

    def slice0(self):
        """The same range as a 0-based Python slice."""
        return slice(self.start - 1, self.stop)

    def __repr__(self):
        return f"{self.start}:{self.stop}"


def unique(seq):
    """Julia `unique`: distinct values in order of first appearance."""
    seen, out = set(), []
    for v in seq:
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


def is_contiguous(vec):
    """src/util.jl:140-155: every distinct value occupies one consecutive block."""
    past = set()
    vec = list(vec)
    for i in range(len(vec) - 1):
        if vec[i + 1] in past:
            return False
        if vec[i + 1] != vec[i]:
            past.add(vec[i])
    return True


def value_to_idx(values):
    """src/util.jl:158-165 (1-based positions; later duplicates win)."""
    return {v: i + 1 for i, v in enumerate(values)}


def keymatch(l_keys, r_keys):
    """src/util.jl:168-184: positions (1-based) of the keys of `l_keys` that occur in `r_keys`, and where."""
    r_idx_of = value_to_idx(r_keys)
    l_idx, r_idx = [], []
    for i, lk in enumerate(l_keys):
        if lk in r_idx_of:
            l_idx.append(i + 1)
            r_idx.append(r_idx_of[lk])
    return l_idx, r_idx


def ids_to_ranges(id_vec):
    """src/util.jl:187-197: one UnitRange per distinct id, in order of first appearance."""
    id_vec = list(id_vec)
    if not is_contiguous(id_vec):
        raise AssertionError("IDs in id_vec need to appear in contiguous chunks.")
    first, last = {}, {}
    for i, v in enumerate(id_vec):
        first.setdefault(v, i + 1)
        last[v] = i + 1
    return [UnitRange(first[u], last[u]) for u in unique(id_vec)]


def ids_to_ind_mat(id_vec):
    """src/util.jl:200-210: Bool indicator matrix, one column per distinct id (order of first appearance)."""
    id_vec = list(id_vec)
    unq = unique(id_vec)
    col = {u: j for j, u in enumerate(unq)}
    mat = np.zeros((len(id_vec), len(unq)), dtype=bool)
    for i, v in enumerate(id_vec):
        mat[i, col[v]] = True
    return mat


def ids_to_batch_index(id_vec):
    """The one-hot matrix of ids_to_ind_mat as the 0-based column index of each row (what the C ABI takes:
    rowval-1 of the CSC row_batches matrix, src/util.jl:588-593) plus the distinct ids."""
    id_vec = list(id_vec)
    unq = unique(id_vec)
    col = {u: j for j, u in enumerate(unq)}
    return np.array([col[v] for v in id_vec], dtype=np.int32), unq


def subset_ranges(ranges, rng):
    """src/util.jl:214-258.  `ranges` sorted and non-overlapping; returns (new_ranges, r_min_idx, r_max_idx) with
    1-based indices into `ranges`; (.., 1, 0) when nothing intersects."""
    ranges = [UnitRange(*r) for r in ranges]
    rng = UnitRange(*rng)
    if len(ranges) == 0:
        return [], 1, 0
    r_min = max(rng.start, ranges[0].start)
    r_max = min(rng.stop, ranges[-1].stop)
    if r_min > r_max:
        return [], 1, 0
    starts = [r.start for r in ranges]
    # searchsorted(starts, r_min).stop == number of starts <= r_min
    r_min_idx = sum(1 for s in starts if s <= r_min)
    if r_min > ranges[r_min_idx - 1].stop:
        r_min_idx += 1
        if r_min_idx > len(ranges):
            return [], 1, 0
        r_min = ranges[r_min_idx - 1].start
    stops = [r.stop for r in ranges]
    # searchsorted(stops, r_max).start == 1 + number of stops < r_max
    r_max_idx = 1 + sum(1 for s in stops if s < r_max)
    if r_max < ranges[r_max_idx - 1].start:
        r_max_idx -= 1
        if r_max_idx < 1:
            return [], 1, 0
        r_max = ranges[r_max_idx - 1].stop
    if r_min_idx > r_max_idx:
        return [], 1, 0
    new = list(ranges[r_min_idx - 1:r_max_idx])
    new[0] = UnitRange(r_min, new[0].stop)
    new[-1] = UnitRange(new[-1].start, r_max)
    return new, r_min_idx, r_max_idx


def shift_range(rng, delta):
    """src/util.jl:260-262."""
    return UnitRange(rng.start + delta, rng.stop + delta)


def nansum(x):
    x = np.asarray(x, dtype=float)
    return float(np.sum(x[~np.isnan(x)]))


def nanmean(x):
    x = np.asarray(x, dtype=float)
    return float(np.mean(x[~np.isnan(x)]))


def nanvar(x):
    x = np.asarray(x, dtype=float)
    return float(np.var(x[~np.isnan(x)], ddof=1))


def featuresets_to_dense(feature_ids, feature_sets):
    """src/util.jl:453-477 featuresets_to_csc, as a dense L x N float32 matrix: S[l, j] = 1/sqrt(|set l|) for
    feature j in set l (sets may contain ids that are not in feature_ids only if the caller filtered them)."""
    f_to_j = value_to_idx(feature_ids)
    S = np.zeros((len(feature_sets), len(feature_ids)), dtype=np.float32)
    for i, fs in enumerate(feature_sets):
        fs = list(fs)
        scale = np.float32(1.0 / np.sqrt(len(fs)))
        for f in fs:
            S[i, f_to_j[f] - 1] = scale
    return S
