"""BatchArray (src/batch_array.jl:5-117): structure and index bookkeeping on the host.

The numeric operators (+, *, exp and their pull-backs, ba_map) are NOT implemented here: they are part of the
fused HIP data pass (csrc/pmf_fused.hip.inc, k_layer_grad in csrc/pmf_hip.hip).  This class only carries what the
C ABI needs: column ranges, the one-hot row-batch assignment as an index vector, and the nb x N_v value matrices."""
import numpy as np

from .util import UnitRange, ids_to_batch_index, ids_to_ranges, shift_range, subset_ranges, unique


class BatchArray:
    def __init__(self, col_ranges, col_range_ids, row_idx, row_batches, row_batch_ids, values):
        self.col_ranges = tuple(col_ranges)          # UnitRanges (1-based inclusive), possibly with gaps
        self.col_range_ids = list(col_range_ids)
        self.row_idx = row_idx                       # None = all rows; else 1-based row indices (row_selector)
        self.row_batches = tuple(row_batches)        # per range: int32 batch index (0-based) of every (selected) row
        self.row_batch_ids = tuple(row_batch_ids)    # per range: names of the batches
        self.values = tuple(values)                  # per range: (n_batches x len(range)) array

    @classmethod
    def from_views(cls, feature_views, row_batch_dict, value_dicts):
        """src/batch_array.jl:47-79.  Only views that are keys of `row_batch_dict` are kept."""
        unq_views = unique(feature_views)
        keep = [i for i, v in enumerate(unq_views) if v in row_batch_dict]
        kept_views = [unq_views[i] for i in keep]
        row_batch_ids = [list(row_batch_dict[b]) for b in kept_views]
        col_ranges = ids_to_ranges(feature_views)
        kept_ranges = [col_ranges[i] for i in keep]
        kept_vd = [value_dicts[i] for i in keep]
        row_batches, unq_ids, values = [], [], []
        for rb, cr, vd in zip(row_batch_ids, kept_ranges, kept_vd):
            idx, unq = ids_to_batch_index(rb)
            row_batches.append(idx)
            unq_ids.append(unq)
            v = np.zeros((len(unq), len(cr)), dtype=np.float64)   # reference: zeros(...) = Float64 on CPU (Q8)
            for i, b in enumerate(unq):
                v[i, :] = np.asarray(vd[b] if b in vd else vd[str(b)], dtype=np.float64)
            values.append(v)
        return cls(kept_ranges, kept_views, None, row_batches, unq_ids, values)

    @property
    def n_rows(self):
        return len(self.row_batches[0]) if self.row_batches else 0

    def row_batches_dense(self):
        """The Bool one-hot matrices of the reference (for tests)."""
        out = []
        for idx, ids in zip(self.row_batches, self.row_batch_ids):
            m = np.zeros((len(idx), len(ids)), dtype=bool)
            m[np.arange(len(idx)), idx] = True
            out.append(m)
        return tuple(out)

    def row_selector_dense(self, n_total):
        """The reference's sparse row_selector as a dense 0/1 matrix (for tests)."""
        rows = np.arange(1, n_total + 1) if self.row_idx is None else np.asarray(self.row_idx)
        m = np.zeros((len(rows), n_total), dtype=int)
        m[np.arange(len(rows)), rows - 1] = 1
        return m

    def view(self, idx1, idx2):
        """src/batch_array.jl:83-106: rows `idx1` (1-based indices, UnitRange or None/'all'), columns UnitRange."""
        idx2 = UnitRange(*idx2)
        new_ranges, r_min, r_max = subset_ranges(list(self.col_ranges), idx2)
        shifted = [shift_range(r, 1 - idx2.start) for r in new_ranges]
        sel = slice(r_min - 1, r_max)
        if idx1 is None:
            rows = None
        elif isinstance(idx1, (tuple, UnitRange)) and len(idx1) == 2 and not isinstance(idx1, list):
            rows = np.arange(idx1[0], idx1[1] + 1)
        else:
            rows = np.asarray(idx1, dtype=np.int64)
        if rows is None:
            new_rb = list(self.row_batches[sel])
            new_row_idx = self.row_idx
        else:
            new_rb = [rb[rows - 1] for rb in self.row_batches[sel]]
            new_row_idx = rows if self.row_idx is None else np.asarray(self.row_idx)[rows - 1]
        old_ranges = self.col_ranges[sel]
        new_vals = []
        for v, old, new in zip(self.values[sel], old_ranges, new_ranges):
            lo = new.start - old.start
            new_vals.append(v[:, lo:lo + len(new)])     # a numpy view, like Julia's view(a, :, cr)
        return BatchArray(shifted, self.col_range_ids[sel], new_row_idx, new_rb, self.row_batch_ids[sel], new_vals)

    def zero(self):
        """src/batch_array.jl:109-117."""
        return BatchArray(self.col_ranges, list(self.col_range_ids), self.row_idx,
                          [rb.copy() for rb in self.row_batches], self.row_batch_ids,
                          [np.zeros_like(v) for v in self.values])

    def copy(self):
        return BatchArray(self.col_ranges, list(self.col_range_ids), self.row_idx,
                          [rb.copy() for rb in self.row_batches], self.row_batch_ids,
                          [np.array(v, copy=True) for v in self.values])
