"""reweight_eb! of the column-layer regularizers (ColParamReg regularizers.jl:490-497, BatchArrayReg :818-838,
SequenceReg :928-932), called by basic_fit_reg_weight_eb! at src/fit.jl:712.  Hand-computed expectations (CPU only)."""
import numpy as np


def test_colparamreg_reweight_eb(pkg):
    R = pkg.regularizers
    views = [1] * 4 + [2] * 3
    reg = R.ColParamReg(views, weight=1.0)
    v = np.array([1.0, 2.0, 3.0, 6.0, -1.0, 0.0, 1.0])
    pkg.reweight_eb_(reg, v)
    # view 1: mean 3, sample variance 14/3 ; view 2: mean 0, variance 1
    assert np.allclose(reg.centers, [3.0, 0.0])
    assert np.allclose(reg.weights, [0.6 / (0.1 + 0.5 * 14.0 / 3.0), 0.6 / (0.1 + 0.5 * 1.0)], rtol=1e-6)
    pkg.reweight_eb_(reg, v, mixture_p=0.5)
    assert np.allclose(reg.weights, [0.3 / (0.1 + 0.5 * 14.0 / 3.0), 0.3 / 0.6], rtol=1e-6)


def test_batcharrayreg_and_sequence_reweight_eb(pkg):
    R, L = pkg.regularizers, pkg.layers
    M = 6
    views = [1] * 3 + [2] * 1 + [3] * 2
    batch_dict = {1: ["a", "a", "a", "b", "b", "b"], 2: ["c", "c", "d", "d", "e", "e"]}   # view 3 has no batches
    ct = L.construct_model_layers(views, batch_dict)
    ba = ct.unwrapped(4).theta
    ba.values[0][...] = np.array([[1.0, 2.0, 6.0], [0.0, 0.0, 0.0]])     # view 1: 2 batches x 3 columns
    ba.values[1][...] = np.array([[5.0], [7.0], [9.0]])                  # view 2: 3 batches x 1 column
    ct.unwrapped(3).mu[...] = np.arange(6.0)
    sr = R.construct_layer_reg(views, batch_dict, ct, 1.0)
    pkg.reweight_eb_(sr, ct)
    th_reg = sr.regs[3]
    # view 1, batch a: mean 3, var 7 -> 1/7 ; batch b: var 0 -> inf -> 1 + 0.5*3 ; view 2 (one column): var NaN -> 1.5
    assert np.allclose(th_reg.centers[0], [3.0, 0.0]) and np.allclose(th_reg.centers[1], [5.0, 7.0, 9.0])
    assert np.allclose(th_reg.weights[0], [1.0 / 7.0, 2.5]) and np.allclose(th_reg.weights[1], [1.5, 1.5, 1.5])
    mu_reg = sr.regs[2]
    assert np.allclose(mu_reg.centers, [1.0, 3.0, 4.5])
    assert np.allclose(mu_reg.weights[0], 0.6 / (0.1 + 0.5 * 1.0), rtol=1e-6)
    assert np.isnan(mu_reg.weights[1])                                   # var of one element is NaN in Julia too
    assert np.allclose(mu_reg.weights[2], 0.6 / (0.1 + 0.5 * 0.5), rtol=1e-6)
    ld_reg = sr.regs[1]                                                  # logdelta all zero: zero variance everywhere
    assert np.allclose(ld_reg.weights[0], [2.5, 2.5]) and np.allclose(ld_reg.centers[0], [0.0, 0.0])
