#!/usr/bin/env python3
"""Headline benchmark: PDHG iterations/second, ROF-TV denoising, 4096 x 4096 grayscale, fp32.

One "step" = one PDHG iteration (BackendPDHG::PerformIteration, backend_pdhg.cu:313-381) of
matlab/examples/example_rof_primaldual.m on a synthetic 4096^2 image (BASELINE.json configs[1]):
gradient2d + sum_1d('square', 1, f, 10) + sum_norm2(2, false, 'ind_leq0', 1, 1, 1),
pdhg(stepsize='alg2', residual_iter=10, alg2_gamma=0.5), tolerances 0 (never stops early).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...      (no launcher: starts the N ranks itself as a child torch.distributed.run job)
  python bench.py --config c3|c4    the other single-GPU BASELINE configs with the same JSON schema (default c2 = the headline):
      c3  2048 x 2048 x 64 volumetric TV (gradient3d + sum_norm2(3) + sum_1d square), PDHG alg2, residual_iter 10;
          14 floats/voxel/iteration algorithmic, dominant kernel fused_iter3d_x2_kernel (two iterations per launch)
      c4  TV-L1 flow-like 1024^2 (block.sparse W + gradient2d(L = 2), sum_1d abs + sum_norm2(4) abs), ADMM defaults (10 CG
          iterations per outer iteration); a step = one ADMM iteration; the roofline object is that of the dominant kernel of
          the CG round (its compulsory bytes, SURVEY 8d "generic kernels ... with their own compulsory bytes")

N > 1: BASELINE config 5 -- N independent 4096^2 problems (seeds 42..42+N-1), one per GPU, weak
scaling; the 4 residual sums are all-reduced over RCCL every residual iteration so every rank sees
the global stopping criterion.  value = N * K / max-over-ranks(time).

Inputs are resident in HBM before the timed region.  Timed region = K iterations of the loop a caller
of prost.solve runs (Solver::IterateChecked: the stopping test of solver.cu:141-150 after every observable
iteration, i.e. the host waits for the residual sums of every residual iteration); the rate of the bare
iteration loop (Solver::Iterate, no host wait) is reported beside it as `iterate_only_it_per_s`.
Before the W warm-up steps an untimed clock-ramp prelude runs 1500 of the same iterations (~90 ms;
`prelude_iterations`): the part needs tens of milliseconds of load to reach its steady clocks and the
driver's `--steps 20 --warmup 5` would otherwise measure the ramp.  The JSON line carries
  roofline     : dominant kernel = fused_iter2d_x2_kernel, one launch = TWO whole iterations with the
                 iterate in between kept in registers.
                 `achieved` / `frac`: COMPULSORY bytes of the kernel that ran -- every operand it has to read or
                 write once: 7 floats/pixel per two-iteration launch (read x, f, y1, y2; write x, y1, y2), 9 per
                 voxel in 3-D -- / mean launch time measured with HIP events on the solver's stream inside the
                 timed region / 8 TB/s: a fraction in (0, 1] by construction.
                 `traffic` / `frac_hbm_traffic`: the PHYSICAL figure -- HBM bytes per launch from the PMC
                 counters (profiles/traffic_table.json: FETCH_SIZE doubled per MI355X_MICROARCH.md +
                 WRITE_SIZE, separate --pmc passes), looked up by kernel instance, precision, image size and
                 chunk length of the launch that was timed (null if that geometry was never profiled), divided
                 by the same launch time and the peak.  traffic / compulsory = the kernel's over-fetch.
                 `algorithmic_equiv_frac`: SURVEY 8(d)'s two-pass byte model (11 floats/pixel/iteration x the
                 iterations of the launch) over the same time and peak.  It exceeds 1: the launch keeps the
                 iterate between its two iterations on chip, so the two-pass model is no lower bound for it;
                 it says how fast a two-pass implementation would have to stream to keep up.
                 Runs of <= 40 steps stamp EVERY launch (launches_timed >= 8 at the driver's --steps 20); a
                 stamped launch does not overlap its neighbours, which costs such a run ~4 % of `value`
                 (`iterate_only_it_per_s` is measured without stamps).
  cpu_baseline : the CPU oracle (port of the reference path) timed on this host's cores on a bounded
                 sample of the same workload; `reference_build` = the REAL reference's CPU build
                 (oracle/_ref, single-threaded thrust host backend) timed beside the port at 1024^2
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
TRAFFIC_TABLE = os.path.join(ROOT, "profiles", "traffic_table.json")


def traffic_bytes(kernel, size, chunk_cols, dtype="f32", config=None):
    """HBM bytes per launch of `kernel` (name as KernelTimes reports it) at image side `size` with `chunk_cols` columns
    per wavefront, from the PMC passes recorded in profiles/traffic_table.json (FETCH_SIZE doubled per
    MI355X_MICROARCH.md + WRITE_SIZE, KiB; rows without a "dtype" are fp32).  None when this launch geometry was never profiled."""
    try:
        with open(TRAFFIC_TABLE) as fh:
            rows = json.load(fh)["launches"]
    except (OSError, ValueError, KeyError):
        return None, None
    for r in rows:
        if r["kernel"] == kernel and r["size"] == size and r["chunk_cols"] == chunk_cols and r.get("dtype", "f32") == dtype and r.get("config", "c4" if config == "c4" else None) == config:
            return (2 * r["fetch_size_kib"] + r["write_size_kib"]) * 1024, r.get("source")
    return None, None


def validate_line(out):
    """what a reader of the JSON line alone may rely on: no rate called GB/s exceeds the part's HBM peak unless its NAME says it is an
    equivalent figure (a model's bytes over the measured time), and every roofline fraction on compulsory bytes or traffic is <= 1.
    Raises ValueError otherwise (bench.py calls this before it prints; tests/test_bench_harness.py feeds it lines)."""
    peak = HBM_PEAK_GBPS * max(1, int(out.get("n_gpus") or 1))

    def walk(d, where):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, where + k + ".")
                continue
            if not isinstance(v, (int, float)) or isinstance(v, bool):
                continue
            name = where + k
            if "equiv" in k:
                continue
            if (k.endswith("GBps") or (where.startswith("roofline") and k in ("achieved", "achieved_hbm_traffic"))) and v > peak:
                raise ValueError("bench.py: %s = %.1f GB/s exceeds the HBM peak %.0f GB/s and its name does not say 'equiv'" % (name, v, peak))
            if where.startswith("roofline") and k in ("frac", "frac_hbm_traffic") and v > 1.0:
                raise ValueError("bench.py: %s = %.3f is above 1" % (name, v))
    walk(out, "")
    return out


def kernel_kind(kname):
    """-> (family, suffix) of a name KernelTimes emits (backend_pdhg.cpp, KernelTimes): family in {"primal", "dual", "iter", "iter_x2"},
    suffix in {"", "+mid", "+residuals", "+mid+residuals"}.  The family is read from the kernel's own name token
    (fused_<family><2d|3d>[_mc][_x2]_kernel), never from a substring of the whole name: "resi-dual-s" is not a dual pass."""
    base, plus, tail = kname.partition("+")
    suffix = plus + tail
    if suffix not in ("", "+mid", "+residuals", "+mid+residuals"):
        return None, None
    if base.startswith("fused_iter2d_xk_kernel<") and base.endswith(">") and base[len("fused_iter2d_xk_kernel<"):-1].isdigit():
        return ("iter_xk", suffix) if suffix in ("", "+residuals") else (None, None)      # K iterations per launch (tolerance-class arithmetic)
    if not (base.startswith("fused_") and base.endswith("_kernel")):
        return None, None
    core = base[len("fused_"):-len("_kernel")]              # primal2d, dual3d, iter2d, iter2d_mc, iter2d_x2, iter2d_mc_x2, iter3d_x2 ...
    if core in ("primal2d", "primal3d"):
        return ("primal", suffix) if suffix == "" else (None, None)
    if core in ("dual2d", "dual3d"):
        return ("dual", suffix) if suffix == "" else (None, None)
    if core in ("iter2d", "iter3d", "iter2d_mc"):
        return ("iter", suffix) if suffix in ("", "+residuals") else (None, None)
    if core in ("iter2d_x2", "iter3d_x2", "iter2d_mc_x2"):
        return "iter_x2", suffix
    return None, None


def compulsory_floats(kname, volume):
    """values per pixel / voxel a launch of `kname` (as KernelTimes names it) has to move through HBM once: operands read
    + results written ("algorithmic bytes / unit" column).  None for a name this table does not know."""
    g = 3 if volume else 2                                   # gradient components = dual values per pixel
    pair = 2 + 2 * g + 1                                     # read x, f, y ; write x, y : 7 / 9
    family, suffix = kernel_kind(kname)
    if family == "primal":
        return 3 + g                                         # read x, y, f ; write x : 5 / 6
    if family == "dual":
        return 2 + 2 * g                                     # read y, x_new, x_old ; write y : 6 / 8
    if family == "iter":                                     # residual single launch: + y_prev (+ x_prev in 3-D): 9 / 13
        return pair if suffix == "" else pair + g + (1 if volume else 0)
    if family == "iter_x2":                                  # +mid: the iterate in between is stored as well (x, y): 10 / 13
        return pair + (1 + g if "+mid" in suffix else 0)
    if family == "iter_xk":                                  # K iterations per launch: the same 7 values, whatever K
        return pair
    return None


# ------------------------------------------------------------------------------------------------------------------
# the BASELINE configs: problem, backend, metric text, algorithmic bytes, CPU baseline sample
# ------------------------------------------------------------------------------------------------------------------
ZERO_TOL = dict(tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)


def c4_problem(N, warp=False):
    """SURVEY 8(d) C4 (warp=False: W = [diag(Ix) diag(Iy)]) / its gather-type sibling c4w (warp=True: 4 non-zeros per row at displaced
    columns): prost_amd/synthetic.py, tvl1_flow_problem"""
    from prost_amd import synthetic
    return synthetic.tvl1_flow_problem(N, warp)


def make_config(name, size, volume, seed, fp="fp32"):
    """-> dict(prob, backend, metric, workload, units (pixels / voxels of one problem), alg_floats_per_unit (per iteration; None where
    SURVEY 8d defines no per-iteration figure), size_key (traffic-table key), prelude (default clock-ramp iterations))"""
    import prost_amd as prost
    from prost_amd import synthetic
    if name == "c2":
        prob, u, q, f = synthetic.rof_problem(size, size, lmb=LAMBDA, seed=seed)
        return dict(prob=prob, backend=prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA),
                    metric="PDHG iters/sec, ROF-TV %d^2 %s" % (size, fp), units=size * size, alg_floats_per_unit=ALG_FLOATS_PER_PIXEL, size_key=size,
                    workload="ROF-TV denoising %dx%d grayscale (gradient2d + sum_1d square + sum_norm2 ind_leq0), PDHG alg2, residual_iter=10, "
                             "lambda=10; one independent problem per GPU" % (size, size), prelude=1500,
                    tiny=lambda: synthetic.rof_problem(64, 64, lmb=LAMBDA, seed=1)[0])
    if name == "c3":
        nx, ny, L = volume
        prob, u, q, f = synthetic.tv3d_problem(nx, ny, L, lmb=LAMBDA, seed=seed)
        return dict(prob=prob, backend=prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA),
                    metric="PDHG iters/sec, TV-3D %dx%dx%d %s" % (nx, ny, L, fp), units=nx * ny * L, alg_floats_per_unit=14, size_key="%dx%dx%d" % (nx, ny, L),
                    workload="volumetric TV %dx%dx%d (gradient3d + sum_1d square + sum_norm2(3) ind_leq0), PDHG alg2, residual_iter=10, lambda=10; "
                             "one independent problem per GPU" % (nx, ny, L), prelude=60,
                    tiny=lambda: synthetic.tv3d_problem(32, 32, 8, lmb=LAMBDA, seed=1)[0])
    if name == "c4":
        return dict(prob=c4_problem(size), backend=prost.backend.admm(rho0=1), metric="ADMM iters/sec, TV-L1 flow-like %d^2 %s" % (size, fp),
                    units=size * size, alg_floats_per_unit=None, size_key=size,
                    workload="TV-L1 flow-like %dx%d (block.sparse W = [diag(Ix) diag(Iy)] + gradient2d L=2, sum_1d abs + sum_norm2(4) abs), ADMM rho0=1 with the "
                             "reference defaults (cg_max_iter=10, residual_iter=1); a step = one ADMM iteration = one graph projection (CGLS) + two proxes; "
                             "one independent problem per GPU" % (size, size), prelude=100,
                    tiny=lambda: c4_problem(32))
    if name == "c4w":
        return dict(prob=c4_problem(size, True), backend=prost.backend.admm(rho0=1), metric="ADMM iters/sec, TV-L1 optical flow (warp matrix) %d^2 %s" % (size, fp),
                    units=size * size, alg_floats_per_unit=None, size_key=size,
                    workload="TV-L1 optical flow %dx%d (block.sparse warp matrix W: n x 2n, 4 non-zeros per row at columns displaced by a smooth flow of up to 5 "
                             "pixels, general CSR + gradient2d L=2, sum_1d abs + sum_norm2(4) abs), ADMM rho0=1 with the reference's defaults (cg_max_iter=10); "
                             "BASELINE.json configs[3] as worded" % (size, size), prelude=100,
                    tiny=lambda: c4_problem(32, True))
    raise SystemExit("bench.py: unknown --config %r (c2, c3, c4, c4w)" % name)


# C4: values per PIXEL every kernel of an ADMM outer iteration has to move once (n = 2 px primal entries, m = 5 px rows: W's row + 4
# gradient rows; W as CSR: 2 values + 2 column indices + 1 row start per pixel, W^T: 2 + 2 + 2), DESIGN.md section 3.  Indices count as
# values of 4 bytes; with fp64 the vector entries double, the index arrays do not (itemsize-weighted below).
C4_KERNEL_VALUES = {
    # name: (vector values per pixel, 4-byte index words per pixel)
    "cg_pixel_pq_kernel": (16, 0),            # p, s, tau (6), W (2), Sigma on W's rows (1) -> p, q (7); first round of a solve: 12
    "cg_pixel_xrs_kernel": (28, 0),           # r, q (10), Sigma_W (1), x, p, tau (6), W (2) -> r, x, s (9)
    "cg_step_xr2_kernel": (18, 0),            # x, p, tau -> x ; r, q, sigma -> r, t
    "op_stage_kernel<EpiFwdQ>": (14, 3),      # t (2) through W (2 + indices) and the stencil ; sigma (5) -> q (5)
    "op_stage_kernel<EpiAdjS>": (13, 4),      # t (5) through W^T (2 + indices) and the stencil ; x, tau (4) -> s (2)
    "cg_step_p2_kernel": (10, 0),             # p, s, tau -> p, t
}
# the stages of an outer iteration outside the CG rounds (prost_hip_admm_fused_stage / prost_hip_cgls_init_fused), the same way
C4_OUTER_VALUES = {"AdmmPreX": (16, 0), "EpiPreZK": (29, 3), "InitX": (8, 0), "EpiInitRK": (24, 3), "EpiInitSK": (17, 4), "AdmmPostX2": (14, 0),
                   "EpiPostZK": (29, 3), "prox_f": (16, 0), "EpiResZK": (29, 3), "EpiResXK": (15, 4)}


def c4_values(table, name, w_nnz=2):
    """(vector values, index words) per pixel of a stage at a W with `w_nnz` non-zeros per row: the tables above are written for C4's two;
    every stage that applies W or W^T (Epi...K / EpiFwd / EpiAdj: the names with index words) reads w_nnz - 2 more values and as many more
    column indices per pixel"""
    v = table.get(name)
    if v is None:
        return None
    if name.startswith("cg_pixel") and w_nnz != 2:
        # a general CSR W on the two-launch rounds (round 6): its values and column indices + the row starts of W (launch A: one per pixel) or of
        # W^T (launch B: one per primal entry, two per pixel); the operands gathered at displaced pixels are re-reads of vectors counted once
        return (v[0] - 2 + w_nnz, w_nnz + (1 if "pq" in name else 2))
    extra = (w_nnz - 2) if v[1] > 0 else 0
    return (v[0] + extra, v[1] + extra)


def c4_kernel_bytes(kname, n_px, itemsize=4, w_nnz=2):
    """compulsory bytes of one launch of a kernel of the CG round at the C4 shape (generic kernels: every vector the
    kernel has to read or write once, CSR arrays included)"""
    v = c4_values(C4_KERNEL_VALUES, kname, w_nnz)
    return (v[0] * itemsize + v[1] * 4) * n_px if v else None


def c4_iteration_bytes(path, cg_rounds, n_px, itemsize=4, w_nnz=2):
    """compulsory bytes of ONE ADMM outer iteration at the C4 shape: the stages outside the solve + `cg_rounds` CG rounds of the path
    that ran (two launches per round: admm:pixel-op; four: admm:fused-op).  None for a path this table does not describe."""
    words = lambda t: t[0] * itemsize + t[1] * 4
    outer = sum(words(c4_values(C4_OUTER_VALUES, k, w_nnz)) for k in C4_OUTER_VALUES)
    if path == "admm:pixel-op":
        rnd = words(c4_values(C4_KERNEL_VALUES, "cg_pixel_pq_kernel", w_nnz)) + words(c4_values(C4_KERNEL_VALUES, "cg_pixel_xrs_kernel", w_nnz))
        first = rnd - 4 * itemsize                                  # the first launch A of a solve reads no s and writes no p
    elif path == "admm:fused-op":
        rnd = sum(words(c4_values(C4_KERNEL_VALUES, k, w_nnz)) for k in ("op_stage_kernel<EpiFwdQ>", "cg_step_xr2_kernel", "op_stage_kernel<EpiAdjS>", "cg_step_p2_kernel"))
        first = rnd
    else:
        return None
    return int((outer + first + max(int(cg_rounds) - 1, 0) * rnd) * n_px) if cg_rounds and cg_rounds >= 1 else int(outer * n_px)


def cpu_baseline_c3(volume, max_threads, np_dtype=None):
    """oracle (OpenMP port of the reference path) on a CROP of the volume: 256 x 256 x L voxels of the same synthetic data generator, the
    same backend options (scale_steps_operator off: the port's power iteration alone would take a minute; for gradient operators the
    rescale never fires), about 10 s of iterations; reported in voxel-iterations/s and as the equivalent full-volume rate"""
    import numpy as np

    import oracle
    import prost_amd as prost
    from prost_amd import synthetic
    nx, ny, L = volume
    cx, cy = min(nx, 256), min(ny, 256)
    prob, u, q, f = synthetic.tv3d_problem(cx, cy, L, lmb=LAMBDA, seed=42)
    prob.finalize()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA, scale_steps_operator=False)
    opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    oracle.set_num_threads(min(64, max_threads))
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, opts, np_dtype or np.float32)
    s.initialize()
    threads, rates = cpu_probe_threads(s, max_threads)
    iters, t0 = 0, time.time()
    while True:
        s.iterate(5)
        iters += 5
        el = time.time() - t0
        if el > 8.0 or iters >= 2000:
            break
    oracle.bind_threads(False)
    rate = iters / el
    return {"value": rate * cx * cy * L / (nx * ny * L), "unit": "it/s", "cores": threads, "kind": "port",
            "voxel_iterations_per_s": rate * cx * cy * L, "crop_it_per_s": rate, "threads_probed": {str(k): v for k, v in sorted(rates.items())},
            "sample": "%d PDHG iterations of a %dx%dx%d crop of the volume (same generator, same options), oracle/prost_oracle.cpp with %d pinned OpenMP "
                      "threads (best of a probe up to all logical CPUs, vectors first-touched per thread); value = the crop's voxel-iteration rate divided by "
                      "the %dx%dx%d voxels of the full volume" % (iters, cx, cy, L, threads, nx, ny, L)}


def cpu_baseline_c4(size, backend, max_threads, np_dtype=None, warp=False):
    """oracle ADMM (restatement of backend_admm.cu + cgls.hpp; parity-unpinned by the reference) on the SAME problem,
    about 10 s of outer iterations"""
    import numpy as np

    import oracle
    import prost_amd as prost
    prob = c4_problem(size, warp)
    prob.finalize()
    opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    oracle.set_num_threads(min(64, max_threads))
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, opts, np_dtype or np.float32)
    s.initialize()
    threads, rates = cpu_probe_threads(s, max_threads, iters=1)
    iters, t0 = 0, time.time()
    while True:
        s.iterate(2)
        iters += 2
        el = time.time() - t0
        if el > 8.0 or iters >= 400:
            break
    oracle.bind_threads(False)
    return {"value": iters / el, "unit": "it/s", "cores": threads, "kind": "port", "threads_probed": {str(k): v for k, v in sorted(rates.items())},
            "sample": "%d ADMM iterations (10 CG iterations each) of the same %dx%d fp32 problem, oracle/prost_oracle.cpp with %d pinned OpenMP threads (best of a probe up to all "
                      "logical CPUs, vectors first-touched per thread)" % (iters, size, size, threads)}


def cpu_quota_cores():
    """CPU time this process may use per second, in cores, from the cgroup (v2 cpu.max, v1 cfs_quota / cfs_period); None: unlimited.
    The GPU boxes of this pool show 256 logical CPUs and a quota of 16 cores: more runnable threads than that are throttled."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_thread_candidates(max_threads):
    """thread counts the CPU baselines probe: powers of two from 8 up to the logical CPUs of the host and the host's count itself --
    or, under a cgroup CPU quota of q cores, q / 2, q, 3 q / 2 and 2 q (spinning OpenMP threads beyond the quota are throttled)"""
    quota = cpu_quota_cores()
    if quota and quota < max_threads:
        # never more threads than the quota pays for: a team above it runs in bursts (the throttle stops every thread for the rest of the
        # period once the quota is spent), so a 2-iteration probe can read twice the rate the run then sustains (BENCH_r05: probe 48.9 it/s
        # at 24 threads under a 16-core quota, the 8-second sample 29.3)
        q = max(1, int(quota))
        c = sorted({t for t in (max(1, q // 2), max(1, (3 * q) // 4), q) if 1 <= t <= max_threads})
        return c
    c = [t for t in (8, 16, 32, 64, 128, 256, 512) if t <= max_threads]
    if max_threads not in c:
        c.append(max_threads)
    return c or [1]


def cpu_probe_threads(s, max_threads, iters=4):
    """best thread count for the oracle solver `s`: for every candidate the team is pinned (one thread per physical core first,
    topology order: oracle.bind_threads) and every large vector is first-touched again by the thread that streams it
    (Solver.rehome -- round 4 left all pages on the node of the thread that allocated them, which is what capped the port at
    ~30 it/s on a two-socket host).  -> (threads, {threads: it/s}); the solver is left re-homed and bound for the best count."""
    import oracle
    rates = {}
    for t in cpu_thread_candidates(max_threads):
        oracle.set_num_threads(t)
        oracle.bind_threads(True)
        s.rehome()
        s.iterate(1)
        t0 = time.time()
        s.iterate(iters)
        rates[t] = iters / (time.time() - t0)
        oracle.bind_threads(False)
    best = max(rates, key=rates.get)
    oracle.set_num_threads(best)
    oracle.bind_threads(True)
    s.rehome()
    return best, rates


def cpu_baseline(n_img, max_threads, np_dtype=None):
    """Oracle (CPU restatement of the reference path, OpenMP) on a bounded sample: the same 4096^2
    ROF problem, a handful of iterations (about 10-30 s of CPU work).  The thread count is the best
    of a probe over 8 ... all logical CPUs, threads pinned and the state first-touched per thread (cpu_probe_threads)."""
    import numpy as np

    import oracle
    import prost_amd as prost
    from prost_amd import synthetic
    np_dtype = np_dtype or np.float32
    fp = "fp32" if np_dtype == np.float32 else "fp64"
    prob, u, q, f = synthetic.rof_problem(n_img, n_img, seed=42)
    prob.finalize()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * LAMBDA)
    opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0,
                         tol_abs_primal=0, tol_abs_dual=0)
    oracle.set_num_threads(min(max_threads, 64))          # setup (normest's power iteration) on many cores
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, opts, np_dtype)
    s.initialize()
    best, rates = cpu_probe_threads(s, max_threads)
    # three samples of ~3 s each; the reported value is their MEDIAN, min and max beside it (one sample of 8 s in round 5: two runs on
    # one box differed by 40 %)
    samples, iters = [], 0
    for _ in range(3):
        it_s, t0 = 0, time.time()
        while True:
            s.iterate(10)
            it_s += 10
            el = time.time() - t0
            if el > 3.0 or it_s >= 1000:
                break
        samples.append(it_s / el)
        iters += it_s
    samples.sort()
    oracle.bind_threads(False)
    # the same port on ONE thread (SURVEY 8d asks for both): a few iterations are enough at ~3 it/s
    oracle.set_num_threads(1)
    t1 = time.time()
    s.iterate(3)
    single = 3 / (time.time() - t1)
    out = {"value": samples[1], "value_min": samples[0], "value_max": samples[2], "unit": "it/s", "cores": best, "kind": "port", "single_thread_value": single,
           "threads_probed": {str(k): v for k, v in sorted(rates.items())}, "logical_cpus": max_threads, "cpu_quota_cores": cpu_quota_cores(),
           "sample": "median of 3 samples (%d PDHG iterations in all) of the same %dx%d %s ROF problem, oracle/prost_oracle.cpp, OpenMP with %d threads pinned "
                     "one per core in topology order, every vector first-touched by the thread that streams it (best of a probe over %s threads on %d "
                     "logical CPUs%s)" % (iters, n_img, n_img, fp, best, "/".join(str(t) for t in sorted(rates)), max_threads,
                                          "; the container's cgroup CPU quota is %.0f cores and the probe stays within it" % cpu_quota_cores() if cpu_quota_cores() else "")}
    del s
    rb = reference_build_rate(backend, opts) if np_dtype == np.float32 else None
    out["reference_build"] = rb
    if rb and "value" in rb:
        # the REAL reference's CPU path is single-threaded (thrust host backend); its rate at the bench size, assuming it scales with the
        # pixel count as the port does between the two sizes, and how the port compares with it on one thread at the size both ran
        scale = (rb["size"] / float(n_img)) ** 2
        out["reference_build_scaled_to_bench_size"] = rb["value"] * scale
        out["port_over_reference_one_thread"] = rb["port_one_thread_same_size"] / rb["value"]
        out["note"] = ("one thread, %dx%d: the port runs %.2f x the rate of the reference's own CPU build (the reference instantiates its kernel per "
                       "function and is compiled by clang; the port is g++ -O3 -mavx2 with the same expressions); the reference build scaled by pixel "
                       "count to %dx%d: %.2f it/s on one thread (it has no multi-threaded CPU path)"
                       % (rb["size"], rb["size"], out["port_over_reference_one_thread"], n_img, n_img, out["reference_build_scaled_to_bench_size"]))
    return out


def reference_build_rate(backend, opts, n_ref=1024):
    """The REAL reference's CPU build (oracle/_ref/libprost_ref.so: backend_pdhg.cu, problem.cu, prox functors compiled from
    /root/reference where it lies, thrust host backend = ONE thread) on this host, next to the port on one thread, on the same
    ROF problem at 1024^2 (at 4096^2 its one-virtual-call-per-row setup alone exceeds the time budget of a bench run).
    Rate = (k2 - k1) / (t(k2) - t(k1)) over two runs, which removes the setup.  None where the prebuilt library is absent."""
    import numpy as np

    import oracle
    from oracle import ref
    from prost_amd import synthetic
    if not ref.available():
        return None
    try:
        prob, u, q, f = synthetic.rof_problem(n_ref, n_ref, seed=42)
        prob.finalize()
        R = ref.RefProblem(prob.data, prob.nrows, prob.ncols, np.float32)
        oracle.set_num_threads(1)
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, opts, np.float32)
        s.initialize()
        s.iterate(2)
        # three alternating samples of each (the host is shared: one short sample of either moved the ratio between 0.6 and 1.25), medians
        refs, ports = [], []
        for _ in range(3):
            times = {}
            for k in (2, 32):
                t0 = time.time()
                R.pdhg(backend[1], opts, k)
                times[k] = time.time() - t0
            refs.append(30.0 / max(times[32] - times[2], 1e-9))
            t0 = time.time()
            s.iterate(30)
            ports.append(30.0 / (time.time() - t0))
        refs.sort(); ports.sort()
        return {"value": refs[1], "value_min": refs[0], "value_max": refs[2], "unit": "it/s", "cores": 1, "kind": "reference",
                "port_one_thread_same_size": ports[1], "port_one_thread_min": ports[0], "port_one_thread_max": ports[2], "size": n_ref,
                "sample": "median of 3 samples of 30 PDHG iterations of the fp32 ROF problem at %dx%d (each the difference of a 32- and a 2-iteration "
                          "run: setup removed), alternating with the port on one thread; oracle/_ref/libprost_ref.so = the reference's own "
                          "backend_pdhg.cu / problem.cu / prox functors, thrust host backend" % (n_ref, n_ref)}
    except Exception as e:          # the baseline must never take the bench line down
        return {"error": str(e)}


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run job (this process has
    not touched the GPU and never does), pass the child's rank-0 JSON line through and exit with the child's code.  A process
    that has initialised the GPU is never re-executed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    other = [l for l in p.stdout.splitlines() if not l.startswith("{")]
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    if p.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank child job failed (exit code %d)\n" % (n_ranks, p.returncode))
        if lines:                      # e.g. a run that fell back from RCCL: its line says "value": null and why
            print(lines[-1], flush=True)
        raise SystemExit(p.returncode or 1)
    d = json.loads(lines[-1])
    if d.get("n_gpus") != n_ranks:
        sys.stderr.write("bench.py: asked for %d ranks, the job reports n_gpus = %r\n" % (n_ranks, d.get("n_gpus")))
        raise SystemExit(1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1000 + 5000 iterations = 0.4 s of GPU time.  The part needs ~10 ms of load to reach its steady clocks:
    # with --warmup 20 --steps 500 (a 35 ms run) the pair kernel measures 0.124 ms per launch, in steady state 0.115 ms
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c4w"], help="BASELINE config (default c2 = the headline: ROF-TV 4096^2)")
    ap.add_argument("--steps", type=int, default=None, help="timed iterations (default: 5000 for c2, 200 for c3, 1000 for c4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up iterations (default: 1000 / 20 / 100)")
    ap.add_argument("--size", type=int, default=None, help="image side (default 4096 for c2, 1024 for c4)")
    ap.add_argument("--volume", type=int, nargs=3, default=[2048, 2048, 64], metavar=("NX", "NY", "L"), help="c3: the volume (default 2048 2048 64)")
    ap.add_argument("--prelude-iters", type=int, default=None, help="untimed clock-ramp prelude before the warm-up steps: this many of the same iterations (0: none; default ~100 ms worth)")
    ap.add_argument("--sample-every", type=int, default=0, help="stamp one launch in this many with HIP events (0: chosen from --steps)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="arithmetic type: f32 = the BASELINE metric (default); f64 = the precision the "
                    "reference front end ships with (config.hpp:7)")
    ap.add_argument("--stepsize", default=None, choices=["alg1", "alg2", "goldstein", "boyd"], help="pdhg configs: step-size rule (default alg2, the "
                    "example's; boyd with --residual-iter 1 = the reference's DEFAULT backend options, pdhg.m:4-14)")
    ap.add_argument("--residual-iter", type=int, default=None, help="pdhg configs: residual_iter (default 10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events inside the timed region")
    ap.add_argument("--arithmetic", default="exact", choices=["exact", "fmad"], help="pdhg configs: the arithmetic class of the MAIN leg (`value`); default exact = "
                    "bit for bit with the CPU oracle.  With the default, c2 / c3 in f32 run a second leg with 'fmad' and report it as value_fmad / roofline_fmad")
    ap.add_argument("--no-fmad", action="store_true", help="c2, c3 / f32: skip the second leg that runs the same steps with arithmetic='fmad' (value_fmad, roofline_fmad)")
    ap.add_argument("--no-pair", action="store_true", help="one kernel launch per iteration (allow_pair_kernel = false); not the default configuration")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    dflt = {"c2": (5000, 1000, N_IMG), "c3": (200, 20, None), "c4": (1000, 100, 1024), "c4w": (1000, 100, 1024)}[args.config]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    args.size = dflt[2] if args.size is None else args.size
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)          # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE = %d ranks: n_gpus would not be what was asked for" % (args.gpus, world))
    # PROST_BENCH_FORCE_DIST=1 runs the multi-rank code path (torch.distributed + the native RCCL communicator +
    # residual all-reduce) even with one rank: the only way to exercise it on a 1-GPU box
    multi = world > 1 or os.environ.get("PROST_BENCH_FORCE_DIST", "0") == "1"
    # PROST_BENCH_TRANSPORT=host: every rank on GPU 0, gloo between the processes, the native communicator on the
    # host-callback transport (RCCL refuses two ranks on one device) -- runs the N > 1 code of this file and of the
    # solver on a one-GPU box (tests/test_gpu_multirank.py); not a performance configuration
    host_transport = multi and os.environ.get("PROST_BENCH_TRANSPORT", "rccl") == "host"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch

    import prost_amd as prost
    from prost_amd import synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the prost hot path has no CPU fallback")
    if host_transport:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d, this node has %d (one rank per GPU; PROST_BENCH_TRANSPORT=host puts every "
                         "rank on GPU 0 for tests)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        # PROST_BENCH_FORCE_DIST without a launcher: a one-rank rendezvous on the loopback interface
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
            os.environ.setdefault(k, v)
        # torch.distributed is only the RENDEZVOUS here -- it broadcasts the 128-byte ncclUniqueId, runs the barriers around the
        # timed region and MAX-reduces three timing doubles, all on CPU tensors over gloo.  The ONLY RCCL communicator of a rank
        # is the solver's own (prost.comm_init -> ncclCommInitRank), the one the residual all-reduce runs on and the one
        # config.rccl_nranks counts.  (Round 3 created torch's nccl process group beside it: two communicators per rank.)
        dist.init_process_group(backend="gloo")

    prost.set_gpu(local_rank)
    prost.set_precision("single" if args.dtype == "f32" else "double")
    rccl_fallback = False
    if host_transport:
        prost.comm_init_host(lambda a: dist.all_reduce(torch.from_numpy(a)), world)
    elif multi:
        # RCCL communicator owned by the native solver: rank 0 creates the id, gloo broadcasts it
        ident = torch.zeros(128, dtype=torch.float64)
        if rank == 0:
            ident.copy_(torch.from_numpy(prost.comm_unique_id()))
        dist.broadcast(ident, src=0)
        failure = os.environ.get("PROST_BENCH_INJECT_RCCL_FAILURE")          # tests only: the branch below on a box where RCCL works
        try:
            if failure:
                raise RuntimeError(failure)
            prost.comm_init(ident.numpy(), rank, world)
        except Exception as e:                                                # noqa: BLE001 -- whatever the native call raised is reported below
            failure = str(e)
        failed = torch.tensor([1.0 if failure else 0.0])
        dist.all_reduce(failed)
        if failed.item() > 0:
            # the RCCL communicator could not be set up on this node.  The run goes on over the host-callback transport (gloo, 32
            # bytes per residual check) so that the log still shows where the job stands -- but it is NOT the configuration that
            # was asked for: the JSON line then carries "value": null (the rate goes to "value_without_rccl") and
            # "rccl_nranks": null, and the process exits non-zero.  PROST_BENCH_TRANSPORT=host asks for that transport
            # explicitly (tests on one GPU).
            sys.stderr.write("bench.py: rank %d: native RCCL communicator failed (%s); residual all-reduce through the host-callback "
                             "transport over gloo -- this run does not count as an RCCL run\n" % (rank, failure or "on another rank"))
            prost.comm_destroy()
            prost.comm_init_host(lambda a: dist.all_reduce(torch.from_numpy(a)), world)
            rccl_fallback = True
    comm_info = prost.comm_info() if multi else {"nranks": 0, "transport": "none"}
    if multi and int(comm_info["nranks"]) != world:
        raise SystemExit("bench.py: the communicator counts %d ranks, WORLD_SIZE is %d" % (int(comm_info["nranks"]), world))

    n = args.size
    itemsize = 4 if args.dtype == "f32" else 8
    cfg = make_config(args.config, args.size, tuple(args.volume), 42 + rank, "fp32" if args.dtype == "f32" else "fp64")
    prob, backend = cfg["prob"], cfg["backend"]
    if args.stepsize is not None or args.residual_iter is not None:
        if backend[0] != "pdhg":
            raise SystemExit("bench.py: --stepsize / --residual-iter apply to the pdhg configs")
        if args.stepsize is not None:
            backend[1]["stepsize"] = args.stepsize
        if args.residual_iter is not None:
            backend[1]["residual_iter"] = args.residual_iter
        cfg["workload"] = cfg["workload"].replace("PDHG alg2, residual_iter=10", "PDHG %s, residual_iter=%d" % (backend[1]["stepsize"], backend[1]["residual_iter"]))
    if args.arithmetic != "exact":
        if backend[0] != "pdhg":
            raise SystemExit("bench.py: --arithmetic applies to the pdhg configs")
        backend[1]["arithmetic"] = args.arithmetic
    if os.environ.get("PROST_BENCH_DEVICE_RULES") == "0" and backend[0] == "pdhg":
        backend[1]["allow_device_rules"] = False          # A/B: goldstein / boyd with the rule on the host (a wait per residual iteration)
    if args.no_pair:
        if backend[0] != "pdhg":
            raise SystemExit("bench.py: --no-pair applies to the pdhg configs")
        backend[1]["allow_pair_kernel"] = False
    opts = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0,
                         tol_abs_primal=0, tol_abs_dual=0)
    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # launches stamped with events (hipExtLaunchKernel: a start marker in front of the kernel, a stop event bound to the kernel's own
    # command -- the kernel's duration as rocprofv3 reports it).  The marker costs the chain ~3.4 us per stamped launch
    # (tools/stamp_probe.hip, profiles/r04_stamp_probe.txt; events without the system fence), ~3 % of `value` when every launch of the
    # driver's --steps 20 is stamped -- which it is: 8 + 2 samples, a roofline fraction resting on three cannot resolve the spread
    # between boxes (round-3 review).  Longer runs stamp one launch in four / eight.  (Stop events alone are free but give the launch
    # PERIOD, ~4 us more than the kernel's duration: backend_pdhg.cpp, BeginSample.)
    every = args.sample_every or (1 if args.steps <= 40 else 4 if args.steps <= 160 else 8)
    units = cfg["units"]

    def run_leg(backend):
        """tiny warm-up of the code objects, the solver on the real problem, prelude, W warm-up steps, K timed steps (barrier +
        synchronize on both sides), the bare iteration loop, the state: one measurement of one backend description"""
        # code objects are loaded on a kernel's first launch (milliseconds each): run the same kernel instances once on a
        # tiny problem so that a short --warmup does not pay for that inside or right before the timed region
        tiny = prost.Solver(cfg["tiny"](), backend, opts)
        tiny.iterate(24)
        tiny.destroy()
        solver = prost.Solver(prob, backend, opts)          # uploads the data, allocates the state in HBM
        # untimed clock-ramp prelude on the real problem: a FIXED number of the same iterations (about 90 ms at the headline
        # size; every rank must run the same count -- the residual all-reduces pair up across ranks)
        prelude_iters, t_pre = (cfg["prelude"] if args.prelude_iters is None else args.prelude_iters), time.perf_counter()
        if prelude_iters > 0:
            solver.iterate(prelude_iters)
        prelude_ms = (time.perf_counter() - t_pre) * 1e3
        solver.iterate(args.warmup, checked=True)
        barrier()
        t0 = time.perf_counter()
        info = solver.iterate(args.steps, time_kernels=not args.no_kernel_timing, sample_every=every, checked=True, defer_times=True)
        barrier()
        elapsed = time.perf_counter() - t0
        info["kernels"] = solver.kernel_times()          # event pairs recorded inside the timed region, evaluated after it
        # the bare iteration loop (Solver::Iterate: nobody waits for the residual sums), same K, untimed markers off
        barrier()
        elapsed_iterate = solver.iterate(args.steps)["ms"] * 1e-3
        barrier()
        # the timed region proper: the K iterations between the two stream synchronisations INSIDE the native command
        # (info["ms"]); `elapsed` additionally holds the Python -> C marshalling of the call on both sides (~35 us, 3 % of a
        # 20-step run) and is reported as wall_ms_python_side
        t = torch.tensor([info["ms"] * 1e-3, elapsed_iterate, elapsed], dtype=torch.float64)          # CPU tensor: gloo
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if args.config == "c3":
            # the state is 8 GB at the full size: scalars only, finiteness on three 64 Ki-element windows of x and y read from the device
            st = solver.state(vectors=False)
            seg = min(65536, units)
            probes = [np.asarray(solver.read(v, [0, (ln - seg) // 2, ln - seg], seg)) for v, ln in (("x", units), ("y", 3 * units))]
            finite = bool(all(np.isfinite(p).all() for p in probes))
        else:
            st = solver.state()
            finite = bool(np.isfinite(st["x"]).all() and np.isfinite(st["y"]).all())
            st = {k: v for k, v in st.items() if k not in ("x", "y", "z", "w")}
        solver.destroy()
        return {"elapsed": float(t[0].item()), "elapsed_iterate": float(t[1].item()), "elapsed_py": float(t[2].item()), "info": info, "st": st,
                "finite": finite, "prelude_iters": prelude_iters, "prelude_ms": prelude_ms}

    leg = run_leg(backend)
    elapsed, elapsed_iterate, elapsed_py, info, st, finite = leg["elapsed"], leg["elapsed_iterate"], leg["elapsed_py"], leg["info"], leg["st"], leg["finite"]
    prelude_iters, prelude_ms = leg["prelude_iters"], leg["prelude_ms"]
    path = st["path"]
    # the tolerance-class leg (headline config, fp32): the same problem, the same loop, `arithmetic="fmad"` (DESIGN.md section 5;
    # iterates within the tolerance of tests/test_gpu_fmad.py of the exact ones).  `value` above stays the exact run.
    leg_fmad = None
    if args.config in ("c2", "c3") and args.dtype == "f32" and not args.no_fmad and not args.no_pair and args.arithmetic == "exact":
        backend_fmad = [backend[0], dict(backend[1])]
        backend_fmad[1]["arithmetic"] = "fmad"
        leg_fmad = run_leg(backend_fmad)

    if rank == 0:
        value = world * args.steps / elapsed
        afu = cfg["alg_floats_per_unit"]
        bytes_per_iter = afu * itemsize * units if afu else None
        out = {
            "metric": cfg["metric"],
            "value": None if rccl_fallback else value,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config,
                       "path": path, "problems": world, "rccl_nranks": int(comm_info["nranks"]) if comm_info["transport"] == "rccl" else None,
                       "comm_nranks": int(comm_info["nranks"]),
                       "residual_allreduce": "host-callback (gloo)" if host_transport else "host-callback (gloo) after the native RCCL communicator failed" if rccl_fallback
                       else "rccl" if multi else "none",
                       "rendezvous": "gloo (CPU tensors: unique id, barriers, timing reduction); the solver's communicator is the only RCCL communicator of a rank" if multi else "none",
                       "stepsize": backend[1].get("stepsize"), "residual_iter": backend[1].get("residual_iter"),
                       "timed_loop": "Solver::IterateChecked = the loop of prost.solve (stopping test after every observable iteration; "
                                     "tolerances 0, so it never fires)"},
            "iterate_only_it_per_s": world * args.steps / elapsed_iterate,
            "wall_ms_python_side": 1e3 * elapsed_py,     # barrier -> command -> barrier as seen from Python
            "prelude_iterations": prelude_iters,
            "prelude_ms": prelude_ms,
            # BASELINE's "achieved HBM GB/s" as SURVEY 8(d) defines it: it/s x the two-pass model's bytes per iteration (738.2 MB at
            # 4096^2 fp32) -- an EQUIVALENT rate (what a two-pass implementation would have to stream), above the 8 TB/s peak for
            # launches that block several iterations in time, hence the name; the physical figures are roofline.achieved (compulsory
            # bytes) and roofline.achieved_hbm_traffic (PMC traffic)
            "two_pass_equiv_GBps": value * bytes_per_iter / 1e9 if bytes_per_iter else None,
            "algorithmic_equiv_roofline_frac": value * bytes_per_iter / 1e9 / (HBM_PEAK_GBPS * world) if bytes_per_iter else None,
            "iterates_finite": finite,
        }
        if rccl_fallback:
            out["value_without_rccl"] = value
            out["error"] = "the native RCCL communicator could not be created; the run used the host-callback transport over gloo and does not count"
        if args.config in ("c4", "c4w"):
            out["cg_iterations_last_solve"] = st.get("cg_iterations")
            # nothing the reference holds pins ADMM / CGLS: backend_admm.cu needs cuBLAS (nrm2, axpy) and cuSPARSE (csrmv) to compile, neither
            # exists here; the oracle's ADMM is a restatement checked by properties only
            out["oracle_pin"] = "unpinned (the reference's backend_admm.cu / cgls.hpp need cuBLAS / cuSPARSE; the CPU oracle's ADMM is a restatement)"
        w_nnz = 4 if args.config == "c4w" else 2

        def roofline_of(kern, value, st):
            """the roofline object of one measurement: dominant kernel = largest share of the timed region (mean launch time x launches)"""
            if not kern:
                return None, None
            kname = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
            k = kern[kname]
            ipl = k["iterations_per_launch"]
            if args.config in ("c4", "c4w"):
                # ADMM: the kernels of the CG round, each against its own compulsory bytes (generic kernels)
                comp_bytes = c4_kernel_bytes(kname, units, itemsize, w_nnz)
                alg_bytes = None
                note = ("ADMM has no per-iteration byte figure in SURVEY 8d; frac = COMPULSORY bytes of the dominant kernel of the CG round (every operand read "
                        "or written once, index arrays included: %s) / its launch time / peak.  compulsory_bytes_per_iteration = the same count over EVERY "
                        "kernel of an outer iteration (the stages outside the solve + the CG rounds of the last solve); frac_iteration = value x those bytes "
                        "/ peak: the whole-iteration figure, launch gaps and the host loop included.  frac_hbm_traffic = the dominant kernel's PMC traffic "
                        "(FETCH_SIZE x 2 + WRITE_SIZE, profiles/traffic_table.json) / its launch time / peak.  At 1024^2 the working set of a solve (~160 MB) "
                        "fits the 256 MB Infinity Cache, so that traffic is not all HBM traffic; at 2048^2 (--size 2048) it streams from HBM." % kname)
            else:
                unit_name = "pixel" if args.config == "c2" else "voxel"
                cf = compulsory_floats(kname, args.config == "c3")
                comp_bytes = cf * itemsize * units if cf else None
                # SURVEY 8(d)'s two-pass model: floats/unit/iteration x the iterations one launch performs
                # (two-pass kernels: the pass's own share, 5 primal / 6 dual of 11 in 2-D, 6 / 8 of 14 in 3-D)
                if ipl:
                    floats = afu * ipl
                elif args.config == "c3":
                    floats = 8 if kernel_kind(kname)[0] == "dual" else 6
                else:
                    floats = DUAL_PASS_FLOATS if kernel_kind(kname)[0] == "dual" else ALG_FLOATS_PER_PIXEL - DUAL_PASS_FLOATS
                alg_bytes = floats * itemsize * units
                note = ("frac = COMPULSORY bytes of the kernel that ran (%s values per %s per launch: every operand read once, every result written once; "
                        "one launch = %s iteration(s), the iterates in between never leave the registers) / launch time / peak: <= 1 by construction.  "
                        "frac_hbm_traffic = the same with the PMC traffic (FETCH_SIZE x 2 + WRITE_SIZE) in place of the compulsory bytes; traffic / compulsory = "
                        "the kernel's over-fetch (warm-up columns of a chunk, halo lanes, cache lines a strip boundary cuts).  algorithmic_equiv_frac = SURVEY "
                        "8d's two-pass model (%d values per %s and iteration x the iterations of the launch) over the same time: above 1 because the two-pass "
                        "model is no lower bound for a launch that blocks several iterations in time.  That the work is done: tests/test_gpu_fullsize.py "
                        "(exact class: bit for bit against the CPU oracle at this very size and launch geometry), tests/test_gpu_fmad.py (tolerance class: "
                        "within its stated bounds of that oracle at this size)." % (cf, unit_name, ipl if ipl else "1/2", afu, unit_name))
            t_s = k["avg_ms"] * 1e-3
            achieved = comp_bytes / 1e9 / t_s if comp_bytes else None
            traffic, traffic_src = traffic_bytes(kname, cfg["size_key"], k["chunk_cols"], args.dtype, args.config if args.config in ("c4", "c4w") else None)
            phys = traffic / 1e9 / t_s if traffic else None
            alg = alg_bytes / 1e9 / t_s if alg_bytes else None
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic,
                    "frac_hbm_traffic": phys / HBM_PEAK_GBPS if phys else None,
                    "achieved_hbm_traffic": phys, "traffic_source": traffic_src,
                    "traffic_over_compulsory": traffic / comp_bytes if traffic and comp_bytes else None,
                    "algorithmic_equiv_frac": alg / HBM_PEAK_GBPS if alg else None,
                    "algorithmic_equiv_GBps": alg,
                    "note": note,
                    "compulsory_bytes_per_launch": comp_bytes, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k["avg_ms"],
                    "launches_timed": k["sampled"], "iterations_per_launch": ipl, "chunk_cols": k["chunk_cols"],
                    "sample_every": every,
                    "all_kernels": {name: {"avg_launch_ms": v["avg_ms"], "launches": v["launches"], "launches_timed": v["sampled"],
                                           "iterations_per_launch": v["iterations_per_launch"], "chunk_cols": v["chunk_cols"],
                                           "compulsory_bytes": c4_kernel_bytes(name, units, itemsize, w_nnz) if args.config in ("c4", "c4w")
                                           else (compulsory_floats(name, args.config == "c3") or 0) * itemsize * units or None}
                                    for name, v in kern.items()}}
            if args.config in ("c4", "c4w"):
                it_bytes = c4_iteration_bytes(st["path"], st.get("cg_iterations") or 0, units, itemsize, w_nnz)
                roof["compulsory_bytes_per_iteration"] = it_bytes
                roof["frac_iteration"] = (value / world) * it_bytes / 1e9 / HBM_PEAK_GBPS if it_bytes else None
                roof["cg_rounds_per_iteration"] = st.get("cg_iterations")
            # whole-job rate x the compulsory bytes per iteration of the launches that ran / peak: <= roofline.frac (launch gaps, residual
            # launches, the host's wait for the sums).  Launches of different length (groups of 4, 4 and 2 per residual period) share
            # one byte count, so the per-iteration figure comes from the launches themselves: bytes x launches / iterations
            total_it = sum(v["launches"] * v["iterations_per_launch"] for v in kern.values())
            total_bytes = sum(v["launches"] * ((compulsory_floats(name, args.config == "c3") or 0) * itemsize * units) for name, v in kern.items()) \
                if args.config not in ("c4", "c4w") else 0
            whole = (value / world) * (total_bytes / total_it) / 1e9 / HBM_PEAK_GBPS if total_it and total_bytes else None
            return roof, whole

        roof, whole = roofline_of(info.get("kernels", {}), value, st)
        if roof:
            out["roofline"] = roof
        if whole:
            out["hbm_roofline_frac"] = whole
        if leg_fmad is not None:
            v_f = world * args.steps / leg_fmad["elapsed"]
            roof_f, whole_f = roofline_of(leg_fmad["info"].get("kernels", {}), v_f, leg_fmad["st"])
            out["value_fmad"] = v_f
            out["ms_per_step_fmad"] = 1e3 * leg_fmad["elapsed"] / args.steps
            out["roofline_fmad"] = roof_f
            out["fmad"] = {"arithmetic": leg_fmad["st"].get("arithmetic"), "path": leg_fmad["st"]["path"],
                           "iterations_per_launch_max": leg_fmad["st"].get("iterations_per_launch_max"),
                           "iterate_only_it_per_s": world * args.steps / leg_fmad["elapsed_iterate"], "iterates_finite": leg_fmad["finite"],
                           "two_pass_equiv_GBps": v_f * bytes_per_iter / 1e9 if bytes_per_iter else None, "hbm_roofline_frac": whole_f,
                           "speedup_over_exact": v_f / value,
                           "tolerance": "tests/test_gpu_fmad.py: one iteration within 2 (x) / 4 (y) ulp at the vector's scale of the exact kernels; "
                                        "iterates within 1e-5 k (relative to the vector's largest entry) of the CPU oracle after k iterations, "
                                        "checked at this size (c2: k = 12 and 102 on the whole image; c3: k = 12 on sub-volumes); a solve to 1e-4 "
                                        "stops within one residual period of the exact one",
                           "note": "the same problem and loop with backend.pdhg(..., arithmetic='fmad'): fused multiply-adds and fp32 reciprocal "
                                   "instructions (what nvcc's default -fmad=true makes of the reference's kernels)%s.  `value` is the "
                                   "exact-arithmetic run, bit for bit with the CPU oracle." % (", up to 4 iterations per launch (kernels_fused_iterk.hip)" if args.config == "c2" else "")}
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            if args.config == "c2":
                out["cpu_baseline"] = cpu_baseline(n, threads, np.float32 if args.dtype == "f32" else np.float64)
            elif args.config == "c3":
                out["cpu_baseline"] = cpu_baseline_c3(tuple(args.volume), threads, np.float32 if args.dtype == "f32" else np.float64)
            else:
                out["cpu_baseline"] = cpu_baseline_c4(args.size, backend, threads, np.float32 if args.dtype == "f32" else np.float64, args.config == "c4w")
    else:
        out = None

    if multi:
        prost.comm_destroy()
        dist.destroy_process_group()
    if out is not None:
        # the ONE JSON line, last thing on stdout: RCCL prints its version banner through C stdio while the communicators are
        # created, which sits in the C buffer until the process exits when stdout is a pipe -- flush it first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(validate_line(out)), flush=True)
    if rccl_fallback:
        raise SystemExit(3)          # every rank: the launcher (and through it a parent bench.py) reports the failure


if __name__ == "__main__":
    main()
