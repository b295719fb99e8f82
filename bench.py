#!/usr/bin/env python3
"""Headline benchmark: PDHG iterations/second, ROF-TV denoising, 4096 x 4096 grayscale, fp32.

One "step" = one PDHG iteration (BackendPDHG::PerformIteration, backend_pdhg.cu:313-381) of
matlab/examples/example_rof_primaldual.m on a synthetic 4096^2 image (BASELINE.json configs[1]):
gradient2d + sum_1d('square', 1, f, 10) + sum_norm2(2, false, 'ind_leq0', 1, 1, 1),
pdhg(stepsize='alg2', residual_iter=10, alg2_gamma=0.5), tolerances 0 (never stops early).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: BASELINE config 5 -- N independent 4096^2 problems (seeds 42..42+N-1), one per GPU, weak
scaling; the 4 residual sums are all-reduced over RCCL every residual iteration so every rank sees
the global stopping criterion.  value = N * K / max-over-ranks(time).

Inputs are resident in HBM before the timed region.  The JSON line carries
  roofline     : dominant kernel = fused_iter2d_x2_kernel, one launch = TWO whole iterations with the
                 iterate in between kept in registers: algorithmic bytes (2 x 11 floats/pixel, SURVEY
                 8d) / mean launch time measured with HIP events on the solver's stream (one launch in
                 eight is sampled); the kernel itself moves 7 floats/pixel per launch, so `frac` can
                 exceed 1 -- it is measured against what the reference's algorithm must move
  cpu_baseline : the CPU oracle (port of the reference path) timed on this host's cores on a bounded
                 sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_IMG = 4096
LAMBDA = 10.0
ALG_FLOATS_PER_PIXEL = 11          # primal pass 5 + dual pass 6
DUAL_PASS_FLOATS = 6
HBM_PEAK_GBPS = 8000.0
# HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE,
# KiB -> bytes), collected in separate rocprofv3 --pmc passes on the same workload: profiles/r01_pmc_*.txt
TRAFFIC_BYTES_PER_LAUNCH = {"fused_iter2d_kernel": (2 * 162173 + 199683) * 1024,         # profiles/r01_pmc_traffic.txt
                            "fused_iter2d_x2_kernel": (2 * 168755 + 196609) * 1024}


def cpu_baseline(n_img, max_threads):
    """Oracle (CPU restatement of the reference path, OpenMP) on a bounded sample: the same 4096^2
    ROF problem, a handful of iterations (about 10-30 s of CPU work).  The thread count is the best
    of a short probe over {8, 16, 32, 64} <= cores: the path is memory-bound and over-subscribing
    the two sockets is slower than 16-32 threads (measured 256 threads: 1.3 it/s, 16 threads: 32 it/s)."""
    import numpy as np

    import oracle
    import prost_amd as prost
    from prost_amd import synthetic
    prob, u, q, f = synthetic.rof_problem(n_img, n_img, seed=42)
    prob.finalize()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA)
    opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0,
                         tol_abs_primal=0, tol_abs_dual=0)
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, opts, np.float32)
    s.initialize()
    best, best_rate = 1, 0.0
    for t in [c for c in (8, 16, 32, 64) if c <= max_threads] or [1]:
        oracle.set_num_threads(t)
        s.iterate(1)
        t0 = time.time()
        s.iterate(2)
        rate = 2 / (time.time() - t0)
        if rate > best_rate:
            best, best_rate = t, rate
    oracle.set_num_threads(best)
    iters, t0 = 0, time.time()
    while True:
        s.iterate(10)
        iters += 10
        el = time.time() - t0
        if el > 10.0 or iters >= 400:
            break
    # the same port on ONE thread (SURVEY 8d asks for both): a few iterations are enough at ~1 it/s
    oracle.set_num_threads(1)
    t1 = time.time()
    s.iterate(3)
    single = 3 / (time.time() - t1)
    return {"value": iters / el, "unit": "it/s", "cores": best, "kind": "port", "single_thread_value": single,
            "sample": "%d PDHG iterations of the same %dx%d fp32 ROF problem, oracle/prost_oracle.cpp, OpenMP with %d threads "
                      "(best of a probe over 8/16/32/64 threads on %d logical cores)" % (iters, n_img, n_img, best, max_threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1000 + 5000 iterations = 0.4 s of GPU time.  The part needs ~10 ms of load to reach its steady clocks:
    # with --warmup 20 --steps 500 (a 35 ms run) the pair kernel measures 0.124 ms per launch, in steady state 0.115 ms
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--size", type=int, default=N_IMG, help="image side (default 4096 = the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events inside the timed region")
    ap.add_argument("--no-pair", action="store_true", help="one kernel launch per iteration (allow_pair_kernel = false); not the default configuration")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PROST_BENCH_FORCE_DIST=1 runs the multi-rank code path (torch.distributed + the native RCCL communicator +
    # residual all-reduce) even with one rank: the only way to exercise it on a 1-GPU box
    multi = world > 1 or os.environ.get("PROST_BENCH_FORCE_DIST", "0") == "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch

    import prost_amd as prost
    from prost_amd import synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the prost hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    prost.set_gpu(local_rank)
    prost.set_precision("single")
    if multi:
        # RCCL communicator owned by the native solver: rank 0 creates the id, torch broadcasts it
        ident = torch.zeros(128, dtype=torch.float64, device="cuda")
        if rank == 0:
            ident.copy_(torch.from_numpy(prost.comm_unique_id()))
        dist.broadcast(ident, src=0)
        prost.comm_init(ident.cpu().numpy(), rank, world)

    n = args.size
    prob, u, q, f = synthetic.rof_problem(n, n, lmb=LAMBDA, seed=42 + rank)
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA)
    if args.no_pair:
        backend[1]["allow_pair_kernel"] = False
    opts = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0,
                         tol_abs_primal=0, tol_abs_dual=0)
    # code objects are loaded on a kernel's first launch (milliseconds each): run the same kernel instances once on a
    # 64 x 64 problem so that a short --warmup does not pay for that inside or right before the timed region
    tiny = prost.Solver(synthetic.rof_problem(64, 64, lmb=LAMBDA, seed=1)[0], backend, opts)
    tiny.iterate(24)
    tiny.destroy()
    solver = prost.Solver(prob, backend, opts)          # uploads f, allocates x/y ping-pong buffers in HBM

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    solver.iterate(args.warmup)
    barrier()
    t0 = time.perf_counter()
    info = solver.iterate(args.steps, time_kernels=not args.no_kernel_timing)
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    st = solver.state()
    path = st["path"]
    finite = bool(np.isfinite(st["x"]).all() and np.isfinite(st["y"]).all())

    if rank == 0:
        value = world * args.steps / elapsed
        bytes_per_iter = ALG_FLOATS_PER_PIXEL * 4 * n * n
        out = {
            "metric": "PDHG iters/sec, ROF-TV %d^2 fp32" % n,
            "value": value,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ROF-TV denoising %dx%d grayscale (gradient2d + sum_1d square + sum_norm2 ind_leq0), "
                                   "PDHG alg2, residual_iter=10, lambda=10; one independent problem per GPU" % (n, n),
                       "path": path, "problems": world, "residual_allreduce": "rccl" if multi else "none"},
            "achieved_hbm_GBps": value * bytes_per_iter / 1e9,
            "hbm_roofline_frac": value * bytes_per_iter / 1e9 / (HBM_PEAK_GBPS * world),
            "iterates_finite": finite,
        }
        kern = info.get("kernels", {})
        if kern:
            # dominant kernel = largest share of the timed region (mean launch time x launches)
            kname = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
            k = kern[kname]
            ipl = k["iterations_per_launch"]
            # algorithmic bytes per launch = SURVEY 8(d)'s 11 floats/pixel/iteration x the iterations one
            # launch performs (two-pass kernels: the pass's own share, 5 primal / 6 dual)
            floats = ALG_FLOATS_PER_PIXEL * ipl if ipl else (DUAL_PASS_FLOATS if "dual" in kname else ALG_FLOATS_PER_PIXEL - DUAL_PASS_FLOATS)
            alg_bytes = floats * 4 * n * n
            achieved = alg_bytes / 1e9 / (k["avg_ms"] * 1e-3)
            out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": TRAFFIC_BYTES_PER_LAUNCH.get(kname),
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k["avg_ms"],
                               "launches_timed": k["sampled"], "iterations_per_launch": ipl,
                               "kernel_moves_bytes_per_launch": 7 * 4 * n * n if ipl else None,
                               "all_kernels": {name: {"avg_launch_ms": v["avg_ms"], "launches": v["launches"], "iterations_per_launch": v["iterations_per_launch"]}
                                               for name, v in kern.items()}}
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(n, threads)
    else:
        out = None

    solver.destroy()
    if multi:
        prost.comm_destroy()
        dist.destroy_process_group()
    if out is not None:
        # the ONE JSON line, last thing on stdout (RCCL prints its version banner while the communicators are created)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
