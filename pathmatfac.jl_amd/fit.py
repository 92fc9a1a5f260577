"""Fit orchestration above the drop-in boundary (src/fit.jl): `mf_fit!`, `mf_fit_adapt_lr!` and the stage
drivers that only need the gradient-descent loop.  Trailing underscore = the reference's `!`."""
import time

import numpy as np

from . import matfac as MF
from .layers import freeze_layer_, unfreeze_layer_
from .optimizers import AdaGrad

FIT_START_TIME = time.time()


def history_(hist, d=None, **kw):
    """history! (src/util.jl:607-622): the kwargs form stores elapsed seconds, the Dict form absolute time (Q7)."""
    if hist is None:
        return
    if d is None:
        d = {str(k): v for k, v in kw.items()}
        d["time"] = time.time() - FIT_START_TIME
    else:
        d = dict(d)
        d.update({str(k): v for k, v in kw.items()})
        d["time"] = time.time()
    hist.append(d)


def mf_fit_(model, scale_column_losses=False, update_X=False, update_Y=False, update_row_layers=False,
            update_col_layers=False, update_noise_models=True, reg_relative_weighting=False, update_X_reg=False,
            update_Y_reg=False, update_row_layers_reg=False, update_col_layers_reg=False, keep_history=True,
            device=0, **kwargs):
    """mf_fit! (src/fit.jl:9-38): the drop-in boundary.  MF.fit! is replaced by the HIP library."""
    ctx = model.device_context(device)
    return MF.fit_(model.matfac, ctx, update_X=update_X, update_Y=update_Y, update_col_layers=update_col_layers,
                   keep_history=keep_history, **kwargs)


def construct_optimizer(model, lr):
    """src/fit.jl:41-43."""
    return AdaGrad(lr)


def mf_fit_adapt_lr_(model, lr=1.0, min_lr=0.001, max_epochs=1000, history=None, keep_history=True, verbosity=1,
                     print_prefix="", **kwargs):
    """mf_fit_adapt_lr! (src/fit.jl:46-75): halve eta on "loss_increase" until eta < min_lr, resuming the epoch count."""
    opt = construct_optimizer(model, lr)
    epoch = 1
    last = None
    while epoch <= max_epochs:
        h = mf_fit_(model, opt=opt, max_epochs=max_epochs, epoch=epoch, keep_history=True,
                    print_prefix=print_prefix, verbosity=verbosity, **kwargs)
        history_(history, h, name=f"mf_fit_lr={opt.eta}")
        last = h
        if h["term_code"] == "loss_increase":
            opt.eta *= np.float32(0.5)
            opt.eta = float(np.float32(opt.eta))
            if opt.eta < min_lr:
                break
            if verbosity > 0:
                print(f"{print_prefix}Resuming with smaller learning rate ({opt.eta})")
            epoch = h["epochs"]
        else:
            break
    return last


def init_theta_(model, capacity=int(10e8), max_epochs=500, lr_theta=1.0, verbosity=1, print_prefix="",
                history=None, **kwargs):
    """init_theta! (src/fit.jl:106-122): freeze layers 1:3, train BatchShift only."""
    ct = model.matfac.col_transform
    freeze_layer_(ct, [1, 2, 3])
    try:
        mf_fit_adapt_lr_(model, lr=lr_theta, update_col_layers=True, capacity=capacity, max_epochs=max_epochs,
                         verbosity=verbosity, print_prefix=print_prefix, history=history)
    finally:
        unfreeze_layer_(ct, [1, 2, 3])
    history_(history, name="init_theta")


def init_factors_(model, verbosity=1, print_prefix="", history=None, lr=1.0, capacity=10 ** 8, max_epochs=1000,
                  init_factors_method="adagrad", rel_tol=1e-5, abs_tol=1e-5, **kwargs):
    """init_factors! (src/fit.jl:249-288), AdaGrad branch (the L-BFGS branch is out of scope)."""
    if init_factors_method != "adagrad":
        raise NotImplementedError("init_factors_method='lbfgs' is out of scope (src/fit_lbfgs.jl)")
    mf_fit_adapt_lr_(model, capacity=capacity, update_X=True, update_Y=True, lr=lr, min_lr=0.05,
                     max_epochs=max_epochs, verbosity=verbosity, print_prefix=print_prefix + "    ",
                     history=history, rel_tol=rel_tol, abs_tol=abs_tol, **kwargs)
    history_(history, name="init_factors")
