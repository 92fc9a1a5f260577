"""Row-sharded multi-GPU fit (no reference counterpart -- the reference is single GPU).

The data matrix is sharded by rows (samples): rank r owns D[lo:hi, :] and X[:, lo:hi]; Y, the column layers,
the noise weights and the Y regularizer are replicated.  One exchange step per epoch:

    epoch_begin            fused data pass on the local rows -> grad X (local), partial grad Y, local loss
    all_reduce(grad Y)     RCCL over xGMI (torch.distributed backend "nccl"), launched asynchronously ...
    epoch_step_local       ... and overlapped with the regularizer + optimizer step of the local X columns
    epoch_step_shared      identical Y step on every rank once grad Y is complete
    all_reduce(loss)       one float64 so that every rank takes the same termination decision (fit.jl:63)

The loop is written against a small step-level interface (the C ABI's pmf_epoch_* calls, _lib.Context) so the
host logic can be exercised on CPU with the gloo backend (tests/test_parallel_gloo.py).
"""
import time

import numpy as np

TERM_MAX_EPOCHS, TERM_LOSS_INCREASE, TERM_ABS_TOL, TERM_REL_TOL, TERM_NONFINITE = (
    "max_epochs", "loss_increase", "abs_tol", "rel_tol", "nonfinite")


def shard_rows(M, world_size, rank):
    """Contiguous, balanced row block [lo, hi) of rank `rank` (0-based, half-open)."""
    base, rem = divmod(int(M), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class _DevPtr:
    """Zero-copy view of a device buffer owned by libpmf_hip.so as a torch tensor (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2}


def grad_tensor(ctx, which):
    """torch tensor aliasing the library's gradient buffer of parameter group `which`."""
    if hasattr(ctx, "grad_tensor"):          # CPU test double
        return ctx.grad_tensor(which)
    import torch
    ptr, n = ctx.grad_device_ptr(which)
    return torch.as_tensor(_DevPtr(ptr, n), device="cuda")


def fit_distributed(ctx, dist=None, group=None, update_X=False, update_Y=False, update_col_layers=False,
                    frozen_layers=0, frozen_regs=0, max_epochs=1000, epoch=1, abs_tol=1e-9, rel_tol=1e-6,
                    tol_max_iters=3, verbosity=0, print_iter=10, loss_device="cuda", **_ignored):
    """MF.fit! (fit.jl:24) over row shards.  `dist` is torch.distributed (already initialised) or None for a
    single process.  Returns the same history dict as Context.fit."""
    import torch
    world = dist.get_world_size(group) if dist is not None else 1
    o = ctx.make_opts(update_X=update_X, update_Y=update_Y, update_col_layers=update_col_layers,
                      frozen_layers=frozen_layers, frozen_regs=frozen_regs, max_epochs=max_epochs, epoch=epoch,
                      abs_tol=abs_tol, rel_tol=rel_tol, tol_max_iters=tol_max_iters)
    shared = []
    if world > 1 and hasattr(ctx, "set_stream") and not getattr(ctx, "_stream_adopted", False):
        # the collectives below are ordered behind torch's current stream: the library's kernels must run there too
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if world > 1:
        if update_Y:
            shared.append(grad_tensor(ctx, "Y"))
        if update_col_layers:
            for name in ("logsigma", "mu", "logdelta", "theta"):
                t = grad_tensor(ctx, name)
                if t.numel() > 0:
                    shared.append(t)
    loss_buf = torch.zeros(1, dtype=torch.float64, device=loss_device if world > 1 else "cpu")
    t0 = time.time()
    term, tol_iters, losses, prev, last_epoch = TERM_MAX_EPOCHS, 0, [], 0.0, epoch - 1
    for ep in range(epoch, max_epochs + 1):
        ctx.epoch_begin(o)
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in shared]
        ctx.epoch_step_local(o)          # overlaps with the all-reduce
        for w in works:
            w.wait()
        ctx.epoch_step_shared(o)
        local, shared_terms = ctx.epoch_loss()
        if world > 1:
            loss_buf[0] = local - shared_terms
            dist.all_reduce(loss_buf, op=dist.ReduceOp.SUM, group=group)
            loss = float(loss_buf.item()) + shared_terms
        else:
            loss = local
        losses.append(loss)
        last_epoch = ep
        if verbosity > 0 and print_iter > 0 and ep % print_iter == 0:
            print(f"({ep}) Loss={loss:.8g}")
        if not np.isfinite(loss):
            term = TERM_NONFINITE
            break
        if len(losses) > 1:
            diff = prev - loss
            if diff < 0:
                term = TERM_LOSS_INCREASE
                break
            which = None
            if abs(diff) < abs_tol:
                which = TERM_ABS_TOL
            elif (abs(diff / loss) if loss != 0 else np.inf) < rel_tol:   # (loss == 0: the C loop's inf, no ZeroDivisionError)
                which = TERM_REL_TOL
            if which is not None:
                tol_iters += 1
                if tol_iters >= tol_max_iters:
                    term = which
                    break
            else:
                tol_iters = 0
        prev = loss
    return {"term_code": term, "epochs": last_epoch, "loss": np.array(losses), "final_loss": losses[-1] if losses else 0.0,
            "seconds": time.time() - t0}
