#!/usr/bin/env python3
"""Writes the golden known-answer fixtures under tests/golden/.

Every value below is DATA transcribed from the reference's own test-suite
(/root/reference/test/runtests.jl; the line each vector comes from is cited next to it):
inputs and the outputs the reference asserts for them.  No reference source text is kept.
Julia is not installed in the build container, so the reference cannot be executed; the
closed-form expected values are the literals the reference tests themselves hold.

Run:  python tests/golden/make_golden.py     (rewrites the *.json files next to this script)
"""
import json
import math
from pathlib import Path

HERE = Path(__file__).resolve().parent


def dump(name, obj):
    with open(HERE / name, "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


def rep_rows(vals, ncol):
    """Julia repeat([a, b], inner=(1, n)) : a length-2 column repeated along columns -> 2 x n."""
    return [[v] * ncol for v in vals]


# --------------------------------------------------------------------------------------
# util_tests  (runtests.jl:19-56)
# --------------------------------------------------------------------------------------
util = {
    "source": "test/runtests.jl:19-56",
    "is_contiguous": [  # :19-21
        {"in": [2, 2, 2, 1, 1, 4, 4, 4], "out": True},
        {"in": ["cat", "cat", "dog", "dog", "fish"], "out": True},
        {"in": [1, 1, 5, 5, 1, 3, 3, 3], "out": False},
    ],
    "ids_to_ranges": [  # :23-25   ranges are [start, stop], 1-based inclusive
        {"in": [2, 2, 2, 1, 1, 4, 4, 4], "out": [[1, 3], [4, 5], [6, 8]]},
        {"in": ["cat", "cat", "dog", "dog", "fish"], "out": [[1, 2], [3, 4], [5, 5]]},
    ],
    "subset_ranges": [  # :28-33   out = (new_ranges, r_min_idx, r_max_idx)
        {"ranges": [[1, 2], [3, 4], [5, 5]], "rng": [2, 5], "out": [[[2, 2], [3, 4], [5, 5]], 1, 3]},
        {"ranges": [[1, 2], [3, 4], [5, 5]], "rng": [2, 8], "out": [[[2, 2], [3, 4], [5, 5]], 1, 3]},
        {"ranges": [[1, 2], [5, 6], [8, 10]], "rng": [1, 3], "out": [[[1, 2]], 1, 1]},
        {"ranges": [[1, 2], [5, 6], [8, 10]], "rng": [5, 8], "out": [[[5, 6], [8, 8]], 2, 3]},
    ],
    "ids_to_ind_mat": {  # :35-48
        "in": [1, 1, 1, 2, 2, 1, 2, 3, 3, 1, 2, 3, 3],
        "out": [[1, 0, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 1, 0], [1, 0, 0], [0, 1, 0],
                [0, 0, 1], [0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 1]],
    },
    "value_to_idx": {"in": ["cat", "dog", "fish", "bird"],  # :50-51 (1-based)
                     "out": {"cat": 1, "dog": 2, "fish": 3, "bird": 4}},
    "keymatch": {"l_keys": ["cat", "dog", "fish", "bird"], "r_keys": ["dog", "bird", "cat"],  # :54-56
                 "l_idx": [1, 2, 4], "r_idx": [3, 1, 2]},
    "nanstats": {"in": [1.0, None, 2.0, None, 3.0], "nansum": 6.0, "nanmean": 2.0, "nanvar": 1.0},  # :59-66 (None = NaN)
}
dump("util_index.json", util)

# --------------------------------------------------------------------------------------
# batch_array_tests  (runtests.jl:123-257)
# --------------------------------------------------------------------------------------
test_mat = [  # :204-208
    [3.14, 3.14, 3.14, 0.0, 0.0, 0.0, -1.0],
    [3.14, 3.14, 3.14, 0.0, 0.0, 0.0, -1.0],
    [3.14, 3.14, 3.14, 0.0, 0.5, 0.5, -1.0],
    [2.7, 2.7, 2.7, 0.0, 0.5, 0.5, -1.0],
    [2.7, 2.7, 2.7, 0.0, 0.5, 0.5, 1.0],
]
other_test_mat = [row[:3] + [1.0] + row[4:] for row in test_mat]  # :220-221  column 4 set to 1
ba = {
    "source": "test/runtests.jl:123-257",
    "inputs": {  # :123-126
        "col_batches": ["cat", "cat", "cat", "bird", "dog", "dog", "fish"],
        "row_batches": {"cat": [1, 1, 1, 2, 2], "dog": [1, 1, 2, 2, 2], "fish": [1, 1, 1, 1, 2]},
        # one value dict per unique column batch, in order of first appearance (cat, bird, dog, fish)
        "values": [{"1": [3.14] * 3, "2": [2.7] * 3}, {}, {"1": [0.0] * 2, "2": [0.5] * 2},
                   {"1": [-1.0], "2": [1.0]}],
        "M": 5, "N": 7,
    },
    "ctor": {  # :139-146
        "col_ranges": [[1, 3], [5, 6], [7, 7]],
        "row_batches": [
            [[1, 0], [1, 0], [1, 0], [0, 1], [0, 1]],
            [[1, 0], [1, 0], [0, 1], [0, 1], [0, 1]],
            [[1, 0], [1, 0], [1, 0], [1, 0], [0, 1]],
        ],
        "values": [rep_rows([3.14, 2.7], 3), rep_rows([0.0, 0.5], 2), rep_rows([-1.0, 1.0], 1)],
    },
    "view_2to4_2to6": {  # :151-163  view(ba, 2:4, 2:6)
        "rows": [2, 4], "cols": [2, 6],
        "col_ranges": [[1, 2], [4, 5]],
        "row_batches": [
            [[1, 0], [1, 0], [0, 1]],   # test_row_batches[1][2:4,:]
            [[1, 0], [0, 1], [0, 1]],   # test_row_batches[2][2:4,:]
        ],
        "row_selector": [[0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0]],
        "values": [rep_rows([3.14, 2.7], 2), rep_rows([0.0, 0.5], 2)],
    },
    "gappy": {  # :178-189
        "col_batches": ["cat", "cat", "cat", "bird", "bird", "bird", "dog", "dog", "fish"],
        "col_ranges": [[1, 3], [7, 8], [9, 9]],
        "empty_view_cols": [4, 6],   # view(gappy_ba, :, 4:6) has no col_ranges / row_batches / values
    },
    "zero_values": [rep_rows([0.0, 0.0], 3), rep_rows([0.0, 0.0], 2), rep_rows([0.0, 0.0], 1)],  # :196-199
    "add": {  # :203-215   Z = zeros(5,7) + ba ; gradient of sum(x+y)
        "A": [[0.0] * 7 for _ in range(5)],
        "Z": test_mat,
        "A_grad": [[1.0] * 7 for _ in range(5)],
        "ba_grad_values": [rep_rows([3.0, 2.0], 3), rep_rows([2.0, 3.0], 2), rep_rows([4.0, 1.0], 1)],
    },
    "mul": {  # :219-230   Z = ones(5,7) * ba ; gradient of sum(x*y)
        "A": [[1.0] * 7 for _ in range(5)],
        "Z": other_test_mat,
        "A_grad": other_test_mat,
        "ba_grad_values": [rep_rows([3.0, 2.0], 3), rep_rows([2.0, 3.0], 2), rep_rows([4.0, 1.0], 1)],
    },
    "exp": {  # :234-240   ones(5,7) * exp(ba) == exp.(test_mat) ; gradient of sum(ones * exp(x))
        "Z": [[math.exp(v) for v in row] for row in test_mat],
        "ba_grad_values": [rep_rows([3.0 * math.exp(3.14), 2.0 * math.exp(2.7)], 3),
                           rep_rows([2.0 * math.exp(0.0), 3.0 * math.exp(0.5)], 2),
                           rep_rows([4.0 * math.exp(-1.0), 1.0 * math.exp(1.0)], 1)],
    },
    "ba_map_identity": {  # :244-250   ba_map(a->a, ba, test_mat)
        "arg": test_mat,
        "out": [[[3 * 3.14] * 3, [2 * 2.7] * 3], [[0.0, 0.0], [1.5, 1.5]], [[-4.0], [1.0]]],
    },
}
dump("batch_array_5x7.json", ba)

# --------------------------------------------------------------------------------------
# BatchArrayReg test fixture (runtests.jl:780-792): same construction, no "bird" column
# --------------------------------------------------------------------------------------
bar = {
    "source": "test/runtests.jl:780-792",
    "col_batches": ["cat", "cat", "cat", "dog", "dog", "fish"],
    "row_batches": {"cat": [1, 1, 1, 2, 2], "dog": [1, 1, 2, 2, 2], "fish": [1, 1, 1, 1, 2]},
    "values": [{"1": [3.14] * 3, "2": [2.7] * 3}, {"1": [0.0] * 2, "2": [0.5] * 2}, {"1": [-1.0], "2": [1.0]}],
    "weight": 1.0,
    # ba_reg(ba) == 0.5*sum(w .* v .* v) ; grad == w .* v       (:787-790)
    "loss": 0.5 * (3 * 3.14 ** 2 + 3 * 2.7 ** 2 + 2 * 0.5 ** 2 + 1.0 + 1.0),
}
dump("batch_array_reg.json", bar)

# --------------------------------------------------------------------------------------
# featureset_ard_tests constructor facts (runtests.jl:814-862)
# --------------------------------------------------------------------------------------
fs = {
    "source": "test/runtests.jl:814-862",
    "N": 40, "K": 10,
    "feature_ids": list(range(1, 41)),
    "feature_views": [1] * 20 + [2] * 20,
    "feature_sets": [[list(range(1, 6)), list(range(6, 11)), list(range(11, 16)), list(range(16, 21))],
                     [list(range(21, 26)), list(range(25, 31)), list(range(31, 36)), list(range(36, 41))]],
    "alpha0": 1.001, "v0": 0.8,
    "col_ranges": [[1, 20], [21, 40]],          # :849
    "featureset_ids": [[1, 2, 3, 4], [1, 2, 3, 4]],  # :850
    "beta_init": 1.001 - 1,                     # :853  (in Float32: Float32(1.001) - 1)
    "A_shapes": [[4, 10], [4, 10]],             # :855-856
    # S[v][l, j - 20(v-1)] = 1/sqrt(|set l|) for j in set l  (:833-839, :860-861)
}
dump("featureset_ard.json", fs)

# --------------------------------------------------------------------------------------
# layers_tests identities (runtests.jl:352-413): shapes and the one literal gradient
# --------------------------------------------------------------------------------------
layers = {
    "source": "test/runtests.jl:352-413",
    "M": 20, "N": 30, "K": 4, "n_col_batches": 2, "n_row_batches": 4,
    # BatchShift gradient of sum(f(x)) wrt theta.values == ones(nb, N/ncb) * (M/nb)   (:412)
    "bshift_theta_grad_value": 5.0,
    # BatchScale: grad wrt input == zeros(M,N) + exp(logdelta)  (:402) ; with logdelta = 0 that is all ones
}
dump("layers.json", layers)
print("golden fixtures written to", HERE)
