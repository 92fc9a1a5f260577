#!/usr/bin/env python3
"""bench.py -- fit! iterations/sec of the PathMatFac gradient-descent loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric): synthetic 200000 x 50000 fp32 data matrix, K = 64, Gaussian loss, group
regularizer on X (32 sample-condition groups), feature-set-ARD regularizer on Y, Adam (north-star; --optimizer adagrad
selects the reference's AdaGrad, same cost) -- one "step" is one
fit! epoch: fused data pass over every local row (forward, masked loss, both gradient GEMMs), regularizer
gradients, optimizer steps of X and Y, deterministic loss reduction.  The rows are sharded over the N GPUs
(strong scaling: total work fixed), grad(Y) is all-reduced with RCCL and overlapped with the local X step.
Inputs are generated on the device and are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = f32 vector peak


def device_for_rank(local_rank, visible):
    """HIP device index of a rank.  A launcher that pins one device per process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    set per rank) leaves ONE visible device, which is then device 0 whatever LOCAL_RANK says; otherwise LOCAL_RANK is the
    device.  A rank that still has no device is a configuration error, reported with the numbers that show it."""
    if visible <= 0:
        raise RuntimeError(f"no HIP device visible to local rank {local_rank}")
    if visible == 1:
        return 0
    if local_rank >= visible:
        raise RuntimeError(f"local rank {local_rank} has no device: only {visible} visible (pin one device per rank or "
                           f"start at most {visible} ranks per node)")
    return local_rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--M", type=int, default=int(os.environ.get("PMF_BENCH_M", 200000)))
    ap.add_argument("--N", type=int, default=int(os.environ.get("PMF_BENCH_N", 50000)))
    ap.add_argument("--K", type=int, default=int(os.environ.get("PMF_BENCH_K", 64)))
    ap.add_argument("--optimizer", default=os.environ.get("PMF_BENCH_OPT", "adam"), choices=["adam", "adagrad"],
                    help="adam = the north-star step; adagrad = the reference's construct_optimizer (src/fit.jl:41-43)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("PMF_BENCH_PRECISION", "f32"), choices=["f32", "bf16x3"],
                    help="products of the fused data pass for the HEADLINE numbers: exact f32 MFMA (default) or the opt-in "
                         "split-bf16 kernel; the other mode is timed afterwards on the same data and reported beside it")
    ap.add_argument("--store", default=os.environ.get("PMF_BENCH_STORE", "f32"), choices=["f32", "bf16"],
                    help="storage type of the device copy of D (bf16: BASELINE configs[4]; read by the split-bf16 pass only)")
    ap.add_argument("--full-model", action="store_true",
                    help="BASELINE configs[2] / configs[4] flavour (SURVEY 8d C3 / C5): 20 %% Bernoulli columns, column scale / shift, "
                         "BatchArray scale / shift on 2 views x 8 row batches, 10 %% missing entries -- on top of the group-reg X + "
                         "feature-set-ARD Y of the headline workload")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        # Started directly with --gpus N: start the N ranks ourselves (one process per GPU), BEFORE anything touches the
        # GPU or imports torch in this process, relay rank 0's JSON line and exit with the launcher's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = max(world, 1)
    if args.gpus != world:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    # stdout carries exactly ONE line, the JSON: everything libraries print there meanwhile (RCCL's version banner at the
    # first collective, for one) goes to stderr
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    # The rank processes never import torch: the GPU is driven through libpmf_hip.so alone (ONE HIP runtime, the system
    # ROCm's, and the RCCL that belongs to it); "torch.cuda.synchronize()" of the contract is ctx.synchronize() here, and
    # the barrier / max over ranks go through the library's communicator (pmf_comm_allreduce).
    os.environ["PMF_NO_TORCH"] = "1"
    import pmf_import
    pkg = pmf_import.load()
    from pathmatfac_jl_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("PMF_FORCE_DIST", "0") == "1"   # exercise the RCCL path with a 1-rank communicator (testing)
    use_dist = world > 1 or force_dist
    M, N, K = args.M, args.N, args.K
    lo, hi = parallel.shard_rows(M, world, rank)
    Ml = hi - lo

    try:
        ctx = pkg.Context(device_for_rank(local_rank, pkg._lib.device_count()))
    except Exception as e:   # a rank that cannot get its device ends the job with its own message
        print(f"[bench rank {rank}] {e}", file=sys.stderr, flush=True)
        sys.exit(3)
    uid_file = None
    if use_dist:
        # ncclUniqueId: created by rank 0, handed to the others through a file (one node; the ranks share their parent,
        # the launcher).  Everything after that goes over RCCL.
        key = os.environ.get("PMF_BENCH_KEY") or (f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_"
                                                  f"{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}")
        uid_file = Path(os.environ.get("TMPDIR", "/tmp")) / f"pmf_bench_uid_{key}"
        if rank == 0:
            uid = pkg._lib.comm_unique_id()
            tmp = uid_file.with_suffix(".tmp")
            tmp.write_bytes(uid)
            os.replace(tmp, uid_file)
        else:
            t_wait = time.time()
            while not (uid_file.exists() and uid_file.stat().st_size == 128):
                if time.time() - t_wait > 600:
                    raise RuntimeError(f"rank {rank}: no unique id from rank 0 at {uid_file}")
                time.sleep(0.05)
            uid = uid_file.read_bytes()
        ctx.comm_init(rank, world, uid)
    if os.environ.get("PMF_BENCH_CHUNKS"):   # column chunks per data pass (default: 1 on one rank, up to 4 with more)
        ctx.comm_set_chunks(int(os.environ["PMF_BENCH_CHUNKS"]))

    def barrier():
        if use_dist:
            ctx.comm_allreduce(np.zeros(1, np.float64))
    seed = 20260104
    rng_y = np.random.default_rng(seed)                       # replicated Y: same stream of numbers on every rank
    Y_true = (rng_y.standard_normal((K, N)) * 0.3).astype(np.float32)
    Y0 = (rng_y.standard_normal((K, N)) * 0.1).astype(np.float32)
    rng_x = np.random.default_rng(seed + 1 + rank)
    X_true = (rng_x.standard_normal((K, Ml)) * 0.3).astype(np.float32)
    X0 = (rng_x.standard_normal((K, Ml)) * 0.1).astype(np.float32)
    ctx.set_data_device(None, Ml, N, store=args.store)        # device-resident, library-owned
    ctx.set_factors(X_true, Y_true)
    if args.full_model:
        # mu_j ~ N(0,1), logsigma_j ~ N(0,0.1); 2 views x 8 row batches (samples grouped by batch over the
        # GLOBAL rows, each rank takes its slice), theta / logdelta = batch centre N(0,0.25^2) + N(0,0.25^2); 20 % Bernoulli
        # columns (first in the column order: model.jl:50-54 sorts columns by distribution); 10 % missing
        rng_m = np.random.default_rng(seed + 5)              # replicated: same numbers on every rank
        ctx.set_col_params((0.1 * rng_m.standard_normal(N)).astype(np.float32), rng_m.standard_normal(N).astype(np.float32))
        nb, half, views = 8, N // 2, []
        for (s1, e1) in ((1, half), (half + 1, N)):
            nv = e1 - s1 + 1
            bor = np.sort(rng_m.integers(0, nb, M)).astype(np.int32)[lo:hi]
            cd, ct = 0.25 * rng_m.standard_normal((nb, 1)), 0.25 * rng_m.standard_normal((nb, 1))
            views.append(dict(start1=s1, stop1=e1, batch_of_row=np.ascontiguousarray(bor),
                              logdelta=(cd + 0.25 * rng_m.standard_normal((nb, nv))).astype(np.float32),
                              theta=(ct + 0.25 * rng_m.standard_normal((nb, nv))).astype(np.float32)))
        ctx.set_batch_views(views)
        nbern = N // 5
        ctx.set_noise([(1, nbern), (nbern + 1, N)], ["bernoulli", "normal"], np.ones(N, np.float32))
        ctx.synth_data(seed=seed + 17 * (rank + 1), noise=0.1, frac_nan=0.1)
    else:
        ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
        ctx.set_batch_views([])
        ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
        ctx.synth_data(seed=seed + 17 * (rank + 1), noise=0.1)    # D = X'Y + 0.1*N(0,1) on the device
    ctx.set_factors(X0, Y0)
    # group regularizer on X: 32 contiguous sample-condition groups over the GLOBAL rows, clipped to the shard
    ng = 32
    edges = np.linspace(0, M, ng + 1).astype(np.int64)
    groups = [(max(int(edges[g]), lo) - lo + 1, min(int(edges[g + 1]), hi) - lo) for g in range(ng)
              if min(int(edges[g + 1]), hi) > max(int(edges[g]), lo)]
    ctx.clear_xreg()
    ctx.add_reg_group("X", groups, np.ones((len(groups), K), np.float32))
    # feature-set ARD on Y: alpha = 1.001, beta = 0.001*(0.8 + A'S) with sparse positive A'S (simulate_params.jl:61-77)
    beta = (0.001 * (0.8 + 2.0 * rng_y.random((K, N)) * (rng_y.random((K, N)) < 0.05))).astype(np.float32)
    ctx.clear_yreg()
    ctx.add_yreg_fsard(np.full(N, 1.001, np.float32), beta)
    lr = 0.05 if args.optimizer == "adagrad" else 0.01

    def run_epochs(first, last):
        """Epochs first..last through pmf_fit (the C loop: data pass, exchange, steps, loss, termination test).  A fit
        that stops on its own (a loss increase) is resumed, as mf_fit_adapt_lr! does, so that exactly last-first+1
        epochs run."""
        ls, ep = [], first
        while ep <= last:
            r = ctx.fit(update_X=True, update_Y=True, epoch=ep, max_epochs=last, abs_tol=0, rel_tol=0)
            ls += list(r["loss"])
            if r["term_code"] == "nonfinite":
                raise RuntimeError("non-finite loss in the benchmark fit")
            ep = r["epochs"] + 1
        return ls

    kern_names = {}   # arithmetic mode -> the kernel family the library actually ran (pmf_debug_last_kernel)

    def timed_run(precision):
        """W untimed + K timed epochs from the same initial factors and a fresh optimizer state."""
        ctx.set_precision(precision)
        ctx.set_factors(X0, Y0)
        ctx.set_optimizer(args.optimizer, lr=lr)
        n_split0 = ctx.get_precision()[1]
        ls = run_epochs(1, args.warmup) if args.warmup > 0 else []
        ctx.kernel_time(reset=True)
        barrier()
        ctx.synchronize()
        t0 = time.perf_counter()
        ls += run_epochs(args.warmup + 1, args.warmup + args.steps)
        ctx.synchronize()
        barrier()
        t = time.perf_counter() - t0
        if use_dist:
            t = float(ctx.comm_allreduce(np.array([t], np.float64), op="max")[0])
        ms, n = ctx.kernel_time()
        kern_names[precision] = {0: "pmf_fused_kernel", 1: "pmf_fused_sb_kernel", 2: "pmf_fused_sb2_kernel", 4: "pmf_fused_sb4_kernel",
                                 8: "pmf_fused_sb8_kernel"}.get(ctx.last_kernel(), "?")
        return t, ms, n, ls, ctx.get_precision()[1] - n_split0

    dt, k_ms, k_n, losses, n_split = timed_run(args.precision)
    other = "bf16x3" if args.precision == "f32" else "f32"
    have_other = args.store == "f32"   # (a bf16-stored matrix is read by the split-bf16 pass only)
    if have_other:
        dt2, k_ms2, k_n2, losses2, n_split2 = timed_run(other)   # the other arithmetic, same data, reported beside the headline
    split_main = args.precision == "bf16x3" and n_split > 0
    cinfo = ctx.comm_info()
    n_chunks = max(1, cinfo["n_chunks"])

    if rank == 0:
        # 6*M*N*K flops per epoch over the local rows; one launch of the fused kernel covers one of the
        # n_chunks column chunks of the pass (1 on a single GPU)
        flops_launch = 6.0 * Ml * N * K / n_chunks
        dsz = 2.0 if args.store == "bf16" else 4.0      # bytes per entry of D in HBM
        achieved = flops_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        out = {
            "metric": "fit_iters_per_sec", "value": args.steps / dt, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("bf16x3 (split-bf16 products, f32 accumulation and elementwise" + (", D stored bf16)" if args.store == "bf16" else ")"))
                     if split_main else "f32",
            "data": "synthetic",
            "config": {"workload": f"fit! epoch on synthetic {M}x{N} matrix (D stored as {args.store}), K={K}, "
                                   + ("20 % Bernoulli + 80 % Gaussian columns, column scale / shift, BatchArray scale / shift on 2 views x 8 "
                                      "row batches, 10 % missing entries, " if args.full_model else "Gaussian loss, ")
                                   + f"group-reg X (32 groups) + featureset-ARD Y, {args.optimizer}; rows sharded over {world} GPU(s)",
                       "full_model": bool(args.full_model),
                       "M": M, "N": N, "K": K, "rows_per_gpu": Ml, "optimizer": args.optimizer, "lr": lr,
                       "parallelism": (f"row-shard x{world}, library RCCL all-reduce of grad(Y) in {n_chunks} column chunks "
                                       f"beside the data pass ({cinfo['reserved_cus']} CUs left to RCCL)") if world > 1 else "single GPU",
                       "rccl_ranks": cinfo["nranks"] if cinfo["transport"] == "rccl" else 0,
                       "collectives_issued": cinfo["n_collectives"], "column_chunks": n_chunks},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "pmf_fused_kernel", "kernel_ms": k_ms, "launches": k_n,
                         "hbm_stream_GBps": dsz * Ml * N / n_chunks / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                         # the co-bound.  Algorithmic HBM bytes of one epoch on this rank = D once
                         # + parameter / optimizer traffic (p, g, state read + p, state written) + the Y-reg beta
                         "hbm_frac": ((dsz * Ml * N + 4.0 * K * (Ml + N) * (7 if args.optimizer == "adam" else 5)
                                       + 4.0 * K * N) / (dt / args.steps)) / 8.0e12,
                         "hbm_peak_GBps": 8000.0},
            "loss_first": losses[0], "loss_last": losses[-1],
        }
        sb_name = kern_names.get("bf16x3", "pmf_fused_sb_kernel" if K <= 32 else ("pmf_fused_sb2_kernel" if K <= 64 else "pmf_fused_sb8_kernel"))
        if split_main:
            # the split-bf16 kernel needs a quarter of the matrix cycles: what bounds it is the D stream
            d_gbps = dsz * Ml * N / n_chunks / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            out["roofline"].update({"bound": "hbm", "achieved": d_gbps, "peak": 8000.0, "unit": "GB/s", "frac": d_gbps / 8000.0,
                                    "kernel": sb_name})
        # the other arithmetic on the same device data, initial factors and optimizer (pmf_set_precision)
        if have_other:
          took = (n_split2 > 0) if other == "bf16x3" else (n_split2 == 0)
          out["other_precision"] = {
            "precision": other, "kernel": sb_name if other == "bf16x3" else "pmf_fused_kernel",
            "kernel_taken": took, "value": args.steps / dt2, "unit": "iters/s", "ms_per_step": dt2 / args.steps * 1e3,
            "kernel_ms": k_ms2, "launches": k_n2, "hbm_stream_GBps": 4.0 * Ml * N / n_chunks / (k_ms2 * 1e-3) / 1e9 if k_ms2 > 0 else 0.0,
            "loss_last": losses2[-1], "loss_last_rel_diff": abs(losses2[-1] - losses[-1]) / abs(losses[-1]),
          }
        # HBM traffic of the dominant kernel: measured offline with rocprofv3 PMC passes (scripts/profile.sh) and
        # committed under profiles/; reported only when it was measured for exactly this workload
        try:
            tr = json.loads((ROOT / "profiles" / "traffic_latest.json").read_text())
            trk = None
            wl = tr["workload"]
            if (wl["M"], wl["N"], wl["K"], wl["n_gpus"]) == (M, N, K, world) and args.store == "f32" and not args.full_model:
                trk = tr["split_bf16"] if split_main else tr
            c4 = tr.get("config4_shard_full" if args.full_model else "config4_shard")
            if c4 and split_main and (c4["workload"]["M"], c4["workload"]["N"], c4["workload"]["K"], c4["workload"]["n_gpus"],
                                      c4["workload"]["store"]) == (M, N, K, world, args.store):
                trk = c4
            if trk is not None:
                out["roofline"]["traffic"] = trk["hbm_read_bytes_per_launch"] + trk["hbm_write_bytes_per_launch"]
                out["roofline"]["traffic_source"] = trk["source"]
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_cpu_baseline:
            # CPU ports of the same epoch on this host's cores, bounded row samples (oracle/cpu_baseline.py): the OpenMP
            # loop nest of the C oracle, and an epoch with the structure of the reference's CPU path (three sgemm per row
            # batch + materialised m x N layer passes).  Reported baselines, not targets.
            from oracle import cpu_baseline as cb
            ncpu = cb.usable_cpus()
            rows = 1500
            t_omp, thr_omp = cb.openmp_port(N, K, rows, 2, seed, args.optimizer, lr)
            t_blas, thr_blas = cb.blas_port(N, K, rows, 2, seed, args.optimizer, lr)
            sg = cb.sgemm_rate(m=rows, n=min(N, 8192), k=K)
            legs = {
                "openmp": {"value": 1.0 / (t_omp * M / rows), "unit": "iters/s", "cores": thr_omp, "kind": "port",
                           "s_per_epoch_on_sample": t_omp, "gflops_on_sample": 6.0 * rows * N * K / t_omp / 1e9},
                "blas": {"value": 1.0 / (t_blas * M / rows), "unit": "iters/s", "cores": thr_blas, "kind": "port",
                         "s_per_epoch_on_sample": t_blas, "gflops_on_sample": 6.0 * rows * N * K / t_blas / 1e9,
                         "numpy_sgemm_gflops_here": sg},
            }
            best = "blas" if t_blas <= t_omp else "openmp"
            # the reference's own CPU-runnable case (configs[0]) whole, and configs[1] on a row sample
            extra = {}
            other_cfgs = (("configs[0] 500x200 K=4", (500, 200, 4, 500, 50)), ("configs[1] 20000x10000 K=32", (20000, 10000, 32, 2000, 2)))
            if os.environ.get("PMF_BENCH_SKIP_OTHER_CPU") == "1":   # (tests of the output contract: keep the run short)
                other_cfgs = ()
            for tag, (m0, n0, k0, r0, ep0) in other_cfgs:
                to, tho = cb.openmp_port(n0, k0, r0, ep0, seed, args.optimizer, lr)
                tb, thb = cb.blas_port(n0, k0, r0, ep0, seed, args.optimizer, lr)
                extra[tag] = {"openmp": {"value": 1.0 / (to * m0 / r0), "cores": tho}, "blas": {"value": 1.0 / (tb * m0 / r0), "cores": thb},
                              "unit": "iters/s", "epochs_timed": ep0, "rows_sampled": r0}
            out["cpu_baseline_other_configs"] = extra
            out["cpu_baseline_legs"] = legs
            out["cpu_baseline"] = {"value": legs[best]["value"], "unit": "iters/s", "cores": legs[best]["cores"], "kind": "port",
                                   "sample": (f"{best} leg of oracle/cpu_baseline.py on {rows} of {M} rows x {N} cols, K={K}, 2 epochs timed "
                                              f"({legs[best]['s_per_epoch_on_sample']:.2f} s/epoch on the sample), scaled by {M}/{rows}; "
                                              f"{ncpu} CPUs usable by this process (affinity / cgroup quota); both legs in cpu_baseline_legs")}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        barrier()
        ctx.comm_destroy()
        if rank == 0 and uid_file is not None:
            uid_file.unlink(missing_ok=True)
    ctx.close()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:   # any rank's failure ends the job non-zero with that rank's message (the launcher stops the others)
        import traceback
        traceback.print_exc()
        print(f"[bench rank {os.environ.get('RANK', '0')}] FAILED: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.exit(1)
