"""Randomised differential run of the PRODUCT paths against the CPU oracle (test infrastructure: needs an MI355X and oracle/).

    python tools/fuzz_parity.py [--cases 300] [--seed 1] [--budget-s 600]

Every case draws a problem shape the fused PDHG kernels recognise (gray / 2-4 channels / volume; ROF, TV-L1 or the inpainting
shape with a 0 / 1 mask), a geometry (widths from 1 column, heights around the wavefront geometry's seams: 4 rows per lane x 62
/ 63 owner lanes, odd heights, heights that are not a multiple of 16 bytes), a step rule, residual_iter, precision and an
iteration count, runs the default path (whatever launch schedule the backend picks: pairs, single launches, two passes,
speculation) and compares x, y, z, w and the step sizes with the oracle BIT FOR BIT.  boyd / goldstein compare residuals with
thresholds; the product accumulates the residual sums in double, the oracle in T, so a mismatch there is re-run with
`allow_fused = False` (the generic kernels, same accumulation as the fused ones): if the two product paths agree with each
other the case is reported as a threshold tie, not as a failure.
Degenerate geometries (one or two columns / rows / planes) put the estimated operator norm more than 0.1 away from 1, so the
tau / sigma rescaling of backend_pdhg.cu:252-262 fires with a norm estimate whose last bits depend on the summation order
(problem.cu:429-478: thrust / cuBLAS reductions in the reference, sequential sums in the oracle, tree reductions here): such
cases start from step sizes that differ in the last place (`setup` in the summary; the elementwise operations have
discontinuities -- a norm2 operation at a zero norm returns zero, l0 and the truncations threshold -- so such a case may
also end far away: counted, not failed).  Half of the cases therefore run with scale_steps_operator = false (no estimate,
tau0 / sigma0 exactly) and are compared bit for bit whatever the geometry.
Prints one line per failing case (with everything needed to reproduce it) and a summary; exit code 1 on failures."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import oracle
import prost_amd as prost
from prost_amd import synthetic

HEIGHTS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 30, 62, 63, 64, 66, 124, 126, 128, 130, 247, 248, 249, 250, 252, 253, 256, 260, 496, 500, 504, 508, 510, 1000, 1028]
NOT_ORACLE = ("allow_fused", "device_cg", "allow_arg_fusion", "cg_graph", "fused_rounds", "allow_speculation", "allow_pair_kernel", "allow_single_kernel", "allow_device_rules", "allow_op_fusion", "pixel_rounds", "residual_sums_in_prox", "group_max")


def draw(rng):
    kind = rng.choice(["gray", "gray", "mc", "vol"])
    c = {"kind": str(kind), "precision": str(rng.choice(["single", "double"])), "step": str(rng.choice(["alg1", "alg2", "alg2", "goldstein", "boyd"])),
         "residual_iter": int(rng.choice([1, 2, 3, 4, 5, 10])), "iters": int(rng.integers(1, 48)), "lmb": float(rng.choice([0.5, 3.0, 10.0, 40.0])),
         "gamma": float(rng.choice([0.05, 0.4, 0.5, 2.0])), "seed": int(rng.integers(0, 1000)), "checked": bool(rng.integers(0, 2))}
    c["scale_steps"] = bool(rng.random() < 0.5)           # False: no operator-norm estimate, the initial steps are exactly tau0 / sigma0
    # a complete prost.solve instead of a fixed number of iterations: callback schedule (solver.cu:130-135, :153, :178), stopping
    # test, warm start (x0 / y0), the read-outs in between
    if rng.random() < 0.3:
        c["solve"] = {"max_iters": int(rng.integers(1, 70)), "num_cback_calls": int(rng.choice([0, 1, 2, 3, 7, 20])),
                      "tol": float(rng.choice([0.0, 0.0, 3e-2, 1e-2, 3e-3])), "warm": bool(rng.random() < 0.3)}
    c["ny"] = int(rng.choice(HEIGHTS)) if rng.random() < 0.8 else int(rng.integers(1, 600))
    if kind == "vol":
        c["ny"] = min(c["ny"], 260)
        c["nx"] = int(rng.integers(1, 40)); c["L"] = int(rng.choice([1, 2, 3, 5, 12, 13, 14, 15, 16, 17, 27, 30]))
        c["data"] = str(rng.choice(["square", "abs"]))
    else:
        c["nx"] = int(rng.integers(1, 80)); c["L"] = 1 if kind == "gray" else int(rng.choice([2, 3, 4]))
        c["data"] = str(rng.choice(["square", "square", "abs", "mask"]))
        # the gradient handed over as prost.block.sparse(spmat_gradient2d(nx, ny, L)) -- the way the reference's examples write it: recognised
        # entry for entry and run on the same kernels with the position-dependent preconditioners of that matrix (other iterates than
        # with block.gradient2d; the oracle runs the matrix as block.sparse)
        c["as_matrix"] = bool(rng.random() < 0.3)
        # the regulariser written on the PRIMAL side (round 4): example_rof_primal.m -- min_problem, sum_norm2('abs') on the constrained
        # variable, the data term on up to four sub-variables -- or example_nonconvex_rof.m -- conjugate(sum_norm2(fn, ...)) on the dual
        # variable; prox_f* is then a Moreau wrap, evaluated inside the one-kernel iterations
        if c["data"] == "square" and rng.random() < 0.3:
            n = c["nx"] * c["ny"] * c["L"]
            cuts = sorted(set(int(v) for v in rng.integers(1, max(2, n), size=int(rng.integers(0, 4))) if 0 < v < n))
            c["primal_form"] = {"cuts": cuts, "fn": "abs"} if rng.random() < 0.6 else {"conj": str(rng.choice(["abs", "huber", "truncquad", "l0"])), "alpha": float(rng.choice([0.3, 2.0, 30.0])),
                                                                                            "beta": float(rng.choice([0.05, 1.0]))}
    return c


def build(c):
    nx, ny, L = c["nx"], c["ny"], c["L"]
    if c["kind"] == "vol":
        return synthetic.tv3d_problem(nx, ny, L, lmb=c["lmb"], seed=c["seed"], data_term=c["data"])[0]
    if c["data"] != "mask" and not c.get("as_matrix") and not c.get("primal_form"):
        return synthetic.rof_problem(nx, ny, L, lmb=c["lmb"], seed=c["seed"], data_term=c["data"])[0]
    f = synthetic.rof_image(nx, ny, L, c["seed"])
    if c.get("primal_form"):
        from reference_matrices import spmat_gradient2d
        pf, n = c["primal_form"], nx * ny * L
        block = prost.block.sparse(spmat_gradient2d(nx, ny, L)) if c.get("as_matrix") else prost.block.gradient2d(nx, ny, L)
        u, g = prost.variable(n), prost.variable(2 * n)
        if "conj" in pf:                                   # example_nonconvex_rof.m:20-45
            prob = prost.min_max_problem([u], [g])
            prob.add_function(u, prost.function.sum_1d("square", 1, f, c["lmb"]))
            prob.add_function(g, prost.function.conjugate(prost.function.sum_norm2(2 * L, False, pf["conj"], 1, 0, 1, 0, 0, pf["alpha"], pf["beta"])))
            prob.add_dual_pair(u, g, block)
            return prob
        bounds = [0] + pf["cuts"] + [n]                    # example_rof_primal.m:15-28
        subs = [prost.sub_variable(u, bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1)] if pf["cuts"] else []
        prob = prost.min_problem([u], [g])
        for i, sv in enumerate(subs):
            prob.add_function(sv, prost.function.sum_1d("square", 1, f[bounds[i]:bounds[i + 1]], c["lmb"], 0, 0))
        if not subs:
            prob.add_function(u, prost.function.sum_1d("square", 1, f, c["lmb"], 0, 0))
        prob.add_function(g, prost.function.sum_norm2(2 * L, False, "abs", 1, 0, 1, 0, 0))
        prob.add_constraint(u, g, block)
        return prob
    mask = (synthetic.hash32(c["seed"] + 7, np.arange(nx * ny * L, dtype=np.uint64)) % 3 != 0).astype(np.float64)
    u, q = prost.variable(nx * ny * L), prost.variable(2 * nx * ny * L)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", mask, f, c["lmb"]) if c["data"] == "mask" else prost.function.sum_1d(c["data"], 1, f, c["lmb"]))
    prob.add_function(q, prost.function.sum_norm2(2 * L, False, "ind_leq0", 1, 1, 1))
    if c.get("as_matrix"):
        from reference_matrices import spmat_gradient2d
        prob.add_dual_pair(u, q, prost.block.sparse(spmat_gradient2d(nx, ny, L)))
    else:
        prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, L))
    return prob


# ---- mode "large": launch geometries at sizes the oracle is too slow for --------------------------------------------------------
def run_large(args, rng):
    """Random LARGE shapes (up to ~40 M elements per vector: several row strips, hundreds of column chunks, planes beyond 2^31 bytes
    are left to tests/test_gpu_fullsize.py), both precisions, 9-14 iterations with residual_iter 2-5 so that every launch kind of the
    schedule runs (plain and residual pairs, single launches, two passes): the fused path against the generic nine-vector path of
    the same solver, every element of x, y and the previous iterate, compared ON THE DEVICE (solver_compare)."""
    t0, done, fails = time.time(), 0, 0
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    while done < args.cases and time.time() - t0 < args.budget_s:
        kind = str(rng.choice(["gray", "gray", "mc", "vol"]))
        c = {"kind": kind, "precision": str(rng.choice(["single", "double"])), "step": str(rng.choice(["alg1", "alg2"])), "residual_iter": int(rng.choice([2, 3, 4, 5])),
             "iters": int(rng.integers(9, 15)), "lmb": float(rng.choice([3.0, 10.0])), "gamma": 0.5, "seed": int(rng.integers(0, 1000)),
             "data": str(rng.choice(["square", "abs"] if kind == "vol" else ["square", "square", "abs", "mask"])), "scale_steps": bool(rng.integers(0, 2))}
        if kind == "vol":
            c["L"] = int(rng.choice([2, 7, 13, 14, 26, 27, 40, 64, 100]))
            budget = 30e6 / c["L"]
            c["ny"] = int(rng.choice([124, 125, 126, 248, 250, 252, 500, 1000, int(rng.integers(60, 1200))]))
            c["nx"] = int(max(8, min(rng.integers(8, 2000), budget // c["ny"])))
        else:
            c["L"] = 1 if kind == "gray" else int(rng.choice([2, 3, 4]))
            budget = 40e6 / c["L"]
            c["ny"] = int(rng.choice([248, 252, 256, 496, 504, 1000, 1008, 1012, 2048, 4096, 4094, 5000, int(rng.integers(200, 6000))]))
            c["nx"] = int(max(8, min(rng.integers(8, 6000), budget // c["ny"])))
        prost.set_precision(c["precision"])
        d = None
        try:
            prob = build(c)
            sol = []
            for fused in (True, False):
                # (scale_steps: both paths estimate the operator norm with the same kernels -- the stencil form of the power iteration --,
                # so the rescaled steps are the same bits on both sides)
                b = prost.backend.pdhg(stepsize=c["step"], residual_iter=c["residual_iter"], alg2_gamma=c["gamma"], scale_steps_operator=c["scale_steps"])
                b[1]["allow_fused"] = fused
                s = prost.Solver(prob, b, o)
                s.iterate(c["iters"] // 2, checked=True); s.iterate(c["iters"] - c["iters"] // 2)
                sol.append(s)
            cmp_ = sol[0].compare(sol[1])
            st = sol[0].state(vectors=False)
            if any(v[0] != 0 for v in cmp_.values()):
                d = "fused vs generic on the device: %s" % cmp_
            elif not (np.isfinite(st["primal_res"]) and np.isfinite(st["dual_res"])):
                d = "non-finite residuals"
            for s in sol:
                s.destroy()
        except Exception as e:                                      # noqa: BLE001
            d = "exception: %s" % e
        done += 1
        if d:
            fails += 1
            print("FAIL %s: %s" % (c, d), flush=True)
    prost.set_precision("double")
    print("fuzz_parity: %d large cases in %.0f s, %d failures" % (done, time.time() - t0, fails))
    return 1 if fails else 0


# ---- mode "sharded": one image over column slabs ---------------------------------------------------------------------------------
def run_sharded(args, rng):
    """prost_amd.distributed.ColumnShardedSolver, slabs in one process: random image / channel count / number of
    slabs / halo width / iteration count; the owned columns of every slab against the oracle's iterates of the WHOLE image, bit for bit."""
    from prost_amd import distributed
    t0, done, fails = time.time(), 0, 0
    opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    while done < args.cases and time.time() - t0 < args.budget_s:
        world, halo, L = int(rng.integers(2, 6)), int(rng.integers(3, 15)), int(rng.choice([1, 1, 2, 3, 4]))
        c = {"world": world, "halo": halo, "L": L, "ny": int(rng.choice(HEIGHTS[8:])), "nx": int(rng.integers(world * (halo + 2), world * (halo + 2) + 120)),
             "iters": int(rng.integers(1, 60)), "precision": str(rng.choice(["single", "double"])), "residual_iter": int(rng.choice([1, 2, 5, 10])),
             "step": str(rng.choice(["alg1", "alg2"])), "seed": int(rng.integers(0, 1000))}
        prost.set_precision(c["precision"])
        dtype = np.float32 if c["precision"] == "single" else np.float64
        nx, ny = c["nx"], c["ny"]
        d = None
        try:
            f = np.asarray(synthetic.rof_image(nx, ny, L, c["seed"])).ravel()

            def make(lo, hi):
                fs = np.concatenate([f[l * nx * ny + lo * ny: l * nx * ny + hi * ny] for l in range(L)])
                return synthetic.rof_problem(hi - lo, ny, L, f=fs)[0]
            backend = prost.backend.pdhg(stepsize=c["step"], residual_iter=c["residual_iter"], alg2_gamma=0.5)
            whole = make(0, nx)
            whole.finalize()
            orc = oracle.Solver(whole.data, whole.nrows, whole.ncols, [backend[0], dict(backend[1], scale_steps_operator=False)], opts, dtype)
            orc.initialize(); orc.iterate(c["iters"])
            ost = orc.state()
            slabs = [distributed.ColumnShardedSolver(make, nx, ny, backend, opts, r, world, halo, transport="local") for r in range(world)]
            for s_ in slabs:
                s_.transport = slabs
            distributed.iterate_group(slabs, c["iters"])
            own = lambda v, planes, c0, c1: np.concatenate([v[k * nx * ny + c0 * ny: k * nx * ny + c1 * ny] for k in planes])
            for s_ in slabs:
                st = s_.owned_state()
                for name, got, want in (("x", st["x"], own(ost["x"], range(L), s_.c0, s_.c1)), ("y1", st["y1"], own(ost["y"], range(L), s_.c0, s_.c1)),
                                        ("y2", st["y2"], own(ost["y"], range(L, 2 * L), s_.c0, s_.c1))):
                    if d is None and not np.array_equal(got, want):
                        d = "slab %d columns [%d, %d): %s differs (max |d| %.3g)" % (s_.rank, s_.c0, s_.c1, name, float(np.abs(got - want).max()))
                s_.destroy()
        except Exception as e:                                      # noqa: BLE001
            d = "exception: %s" % e
        done += 1
        if d:
            fails += 1
            print("FAIL %s: %s" % (c, d), flush=True)
    prost.set_precision("double")
    print("fuzz_parity: %d sharded cases in %.0f s, %d failures" % (done, time.time() - t0, fails))
    return 1 if fails else 0


# ---- mode "generic": random compositions of the operator blocks and elementwise functions ---------------------------------------
EXACT_FUNS = ("zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01", "max_pos0", "l0", "huber", "trunclin", "truncquad")


def draw_generic(rng):
    """one or two primal variables, one to three dual (constrained) ones, every pair coupled by a sparse / gradient / diags / identity /
    zero block or not at all, a function of the sum_1d / sum_norm2 family (conjugated or not, scalar and per-element coefficients)
    on most of them; min-max form through PDHG (exact against the oracle) or constrained form through PDHG / ADMM (ADMM: tolerance)"""
    c = {"kind": "generic", "precision": str(rng.choice(["single", "double"])), "seed": int(rng.integers(0, 10 ** 6)), "iters": int(rng.integers(1, 30)),
         "residual_iter": int(rng.choice([1, 2, 3, 5, 10])), "checked": bool(rng.integers(0, 2)), "gamma": float(rng.choice([0.05, 0.5, 2.0])),
         "form": str(rng.choice(["minmax", "minmax", "min"]))}
    c["scale_steps"] = bool(rng.random() < 0.2)
    c["backend"] = "pdhg" if c["form"] == "minmax" else str(rng.choice(["pdhg", "admm"]))
    if c["backend"] == "pdhg" and rng.random() < 0.2:
        c["solve"] = {"max_iters": int(rng.integers(1, 50)), "num_cback_calls": int(rng.choice([0, 1, 2, 3, 7, 20])),
                      "tol": float(rng.choice([0.0, 0.0, 3e-2, 1e-2])), "warm": bool(rng.random() < 0.3)}
    c["step"] = str(rng.choice(["alg1", "alg2", "goldstein", "boyd"])) if c["backend"] == "pdhg" else "admm"
    return c


def _coeffs(rng, count, fun):
    def form(lo, hi, zero_ok, default):
        r = rng.random()
        if r < 0.4:
            return default
        if r < 0.7:
            return float(rng.uniform(lo, hi))
        return rng.uniform(lo, hi, count)
    a = form(0.5, 2.0, False, 1)
    if rng.random() < 0.1:
        a = (rng.random(count) < 0.7).astype(np.float64)          # the a == 0 branch of elem_operation_1d.hpp:42-44
    b = form(-1.0, 1.0, True, 0)
    cc = form(0.3, 3.0, False, 1)
    d = form(-0.5, 0.5, True, 0)
    e = form(0.0, 1.0, True, 0)
    alpha = float(rng.uniform(0.1, 1.0)) if fun in ("huber", "trunclin", "truncquad") else 0
    beta = float(rng.uniform(0.1, 1.0)) if fun in ("trunclin", "truncquad") else 0
    return a, b, cc, d, e, alpha, beta


def _wrapper_function(rng, count, desc, what):
    """the remaining builders of +prost/+function: projections per group (halfspace, second-order cone, sum-to-one, simplex, quadratic
    epigraph) and the wrappers around any function (transform, permute)"""
    dims = [d for d in (2, 3, 4, 6) if count % d == 0]
    kind = str(rng.choice(["transform", "permute"] + (["halfspace", "soc", "ind_sum", "simplex", "epi_quad"] if dims else [])))
    if desc is not None:
        desc.append("%s: %s" % (what, kind))
    if kind == "transform":
        inner = prost.function.sum_1d(str(rng.choice(("abs", "square", "ind_box01", "huber"))), 1, float(rng.uniform(-1, 1)), float(rng.uniform(0.5, 2)), 0, 0, 0.5)
        return prost.function.transform(inner, float(rng.uniform(0.5, 2)), float(rng.uniform(-1, 1)), float(rng.uniform(0.5, 2)), float(rng.uniform(-0.5, 0.5)),
                                        float(rng.uniform(0, 1)))
    if kind == "permute":
        inner = prost.function.sum_1d(str(rng.choice(("abs", "square", "ind_geq0"))), 1, rng.uniform(-1, 1, count), rng.uniform(0.5, 2, count))
        return prost.function.permute(inner, rng.permutation(count))
    dim = int(rng.choice(dims))
    n, il = count // dim, bool(rng.integers(0, 2))
    if kind == "halfspace":
        return prost.function.sum_ind_halfspace(dim, il, rng.uniform(-1, 1, dim if rng.random() < 0.5 else count), rng.uniform(-1, 1, 1 if rng.random() < 0.5 else n))
    if kind == "soc":
        return prost.function.sum_ind_soc(dim, il, 1.0)
    if kind == "ind_sum":
        return prost.function.sum_ind_sum(dim, il)
    if kind == "simplex":
        return prost.function.sum_ind_simplex(dim, il)
    if desc is not None:
        desc.append("transcendental")                              # helper.hpp:44-105: pow / acos / cos -- the device's and the host's differ in the last place
    return prost.function.sum_ind_epi_quad(dim, il, rng.uniform(0.5, 2, 1 if rng.random() < 0.5 else n), rng.uniform(-1, 1, (dim - 1) * n), rng.uniform(-1, 1, 1 if rng.random() < 0.5 else n))


CONTINUOUS_FUNS = ("zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01", "max_pos0", "huber")


def _function(rng, count, desc=None, what="", continuous=False):
    """continuous: only operations without jumps (no l0 / truncations; norm2 operations with b = d = 0, whose result tends to zero
    with the norm -- elem_operation_norm2.hpp:56-85 returns zero AT a zero norm whatever b and d say)"""
    if not continuous and rng.random() < 0.2:
        return _wrapper_function(rng, count, desc, what)
    fun = str(rng.choice(CONTINUOUS_FUNS if continuous else EXACT_FUNS))
    r = rng.random()
    if desc is not None:
        desc.append("%s: %s %s" % (what, fun, "1d" if r < 0.45 else "norm2"))
    if r < 0.45:
        f = prost.function.sum_1d(fun, *_coeffs(rng, count, fun))
    else:
        dims = [d for d in (1, 2, 3, 4, 6, 7) if count % d == 0]
        dim = int(rng.choice(dims))
        a, b, cc, d, e, alpha, beta = _coeffs(rng, count // dim, fun)
        if continuous:
            b, d = 0, 0
        f = prost.function.sum_norm2(dim, bool(rng.integers(0, 2)), fun, a, b, cc, d, e, alpha, beta)
    conj = rng.random() < 0.3
    if desc is not None and conj:
        desc[-1] += " conjugated"
    return prost.function.conjugate(f) if conj else f


def build_generic(c):
    import scipy.sparse as sp
    rng = np.random.default_rng(c["seed"])
    big = rng.random() < 0.3                                        # (sparse blocks of >= 256 rows can run from row patterns)
    nx, ny, L = int(rng.integers(2, 30 if big else 12)), int(rng.integers(2, 30 if big else 12)), int(rng.choice([1, 1, 2, 3]))
    n = nx * ny * L
    psizes = [n] + ([int(rng.choice([n, int(rng.integers(1, 50))]))] if rng.random() < 0.4 else [])
    rows = []
    for _ in range(int(rng.integers(1, 4))):
        t = str(rng.choice(["grad2d", "grad3d", "sparse", "sparse", "diags", "identity", "kron_id", "id_kron", "stencil"]))
        if t == "stencil":                                          # a stencil written out as a sparse matrix: applied from row patterns (>= 256 rows)
            st = str(rng.choice(["g2", "g3", "band"]))
            rows.append((t, {"g2": 2 * n, "g3": 3 * n, "band": n}[st], st))
            continue
        if t in ("kron_id", "id_kron"):                             # kron(K, I_d) / kron(I_d, K) with K of (mk x n / d): d must divide n
            dl = int(rng.choice([d for d in (1, 2, 3, 4, 5, 6) if n % d == 0]))
            rows.append((t, int(rng.integers(1, 9)) * dl, dl))
        else:
            rows.append((t, {"grad2d": 2 * n, "grad3d": 3 * n, "identity": n}.get(t, int(rng.integers(1, 70)))))
    pv = [prost.variable(k) for k in psizes]
    dv = [prost.variable(r[1]) for r in rows]
    prob = prost.min_max_problem(pv, dv) if c["form"] == "minmax" else prost.min_problem(pv, dv)
    add = prob.add_dual_pair if c["form"] == "minmax" else prob.add_constraint

    # The CSR kernels sum a row sequentially (the order of the oracle) while the mean row length of the matrix -- and, for the
    # adjoint, of its stored transpose -- is at most 6, and with 4 / 16 / 64 cooperating lanes beyond (kernels_linop.hip); cuSPARSE
    # leaves the order open (test_linop_sparse_zero.m compares with 1e-3).  Short rows: exact comparison; long_rows: tolerance.
    c["long_rows"] = bool(rng.random() < 0.15)

    def sparse_block(m, k):
        density = min(1.0, float(rng.uniform(8.0, 40.0)) / max(m, k)) if c["long_rows"] else min(1.0, 4.0 / max(m, k))
        A = sp.random(m, k, density=density, random_state=int(rng.integers(0, 2 ** 31)), format="csc")
        if A.nnz == 0:
            A = sp.csc_matrix(([1.5], ([0], [0])), shape=(m, k))
        return prost.block.sparse(A)
    covered = [False] * len(pv)
    desc = c.setdefault("_desc", [])
    del desc[:]
    desc.append("image %dx%dx%d primal %s rows %s" % (nx, ny, L, psizes, rows))
    for j, row in enumerate(rows):
        t, m = row[0], row[1]
        if t in ("kron_id", "id_kron"):
            dl = row[2]
            Ks = sp.random(m // dl, n // dl, density=min(1.0, 3.0 / max(m // dl, n // dl)), random_state=int(rng.integers(0, 2 ** 31)), format="csc")
            if Ks.nnz == 0:
                Ks = sp.csc_matrix(([0.75], ([0], [0])), shape=(m // dl, n // dl))
            add(pv[0], dv[j], (prost.block.sparse_kron_id if t == "kron_id" else prost.block.id_kron_sparse)(Ks, dl)); covered[0] = True
        elif t == "stencil":
            from reference_matrices import spmat_gradient2d, spmat_gradient3d
            if row[2] == "g2":
                Km = spmat_gradient2d(nx, ny, L)
            elif row[2] == "g3":
                Km = spmat_gradient3d(nx, ny, L)
            else:
                offs = sorted(set(int(v) for v in rng.integers(-ny - 1, ny + 2, int(rng.integers(1, 5)))))
                Km = sp.diags([float(v) for v in rng.uniform(-1, 1, len(offs))], offs, shape=(n, n))
            add(pv[0], dv[j], prost.block.sparse(sp.csc_matrix(Km))); covered[0] = True
        elif t == "grad2d":
            add(pv[0], dv[j], prost.block.gradient2d(nx, ny, L, bool(rng.integers(0, 2)) if L > 1 else False)); covered[0] = True
        elif t == "grad3d":
            add(pv[0], dv[j], prost.block.gradient3d(nx, ny, L, False)); covered[0] = True
        elif t == "identity":
            add(pv[0], dv[j], prost.block.identity(float(rng.choice([1.0, -2.0, 0.5])))); covered[0] = True
        elif t == "diags":
            nd = int(rng.integers(1, 4))
            offs = sorted(set(int(v) for v in rng.integers(-min(m, 3), min(psizes[0], 4), nd)))
            add(pv[0], dv[j], prost.block.diags(m, psizes[0], [float(v) for v in rng.uniform(-2, 2, len(offs))], offs)); covered[0] = True
        else:
            i = int(rng.integers(0, len(pv)))
            add(pv[i], dv[j], sparse_block(m, psizes[i])); covered[i] = True
        for i in range(len(pv)):                                    # further couplings of this row
            if t in ("sparse",) or i == 0:
                continue
            if rng.random() < 0.5:
                add(pv[i], dv[j], sparse_block(m, psizes[i]) if rng.random() < 0.8 else prost.block.zero()); covered[i] = True
    for i in range(len(pv)):
        if not covered[i]:
            add(pv[i], dv[0], sparse_block(rows[0][1], psizes[i]))
    # (stencil matrices built with scipy carry explicit zeros: they count as entries, and a mean row length > 6 of K or of its stored
    # transpose selects the cooperating-lane CSR kernels where the matrix is too small for row patterns)
    for blk in prob.data["linop"]:
        if blk[0] == "sparse" and blk[3][0].nnz > 6 * min(blk[3][0].shape):
            c["long_rows"] = True
    for i, k in enumerate(psizes):
        if rng.random() < 0.85:
            prob.add_function(pv[i], _function(rng, k, desc, "primal %d" % i, c["backend"] == "admm"))
    for j, row in enumerate(rows):
        m = row[1]
        if rng.random() < 0.85:
            prob.add_function(dv[j], _function(rng, m, desc, "dual %d" % j, c["backend"] == "admm"))
    return prob


def solve_opts(c, prob, trace):
    so = c["solve"]
    r = np.random.default_rng(c["seed"] + 99)
    def cb(it, x, y):
        trace.append((int(it), np.array(x, copy=True), np.array(y, copy=True)))
        return False
    o = prost.options(max_iters=so["max_iters"], num_cback_calls=so["num_cback_calls"], verbose=False, tol_rel_primal=so["tol"], tol_rel_dual=so["tol"],
                      tol_abs_primal=so["tol"], tol_abs_dual=so["tol"], interm_cb=cb)
    if so["warm"]:
        o["x0"] = r.uniform(0, 1, prob.ncols); o["y0"] = r.uniform(-0.3, 0.3, prob.nrows)
    return o


def compare_solve(c, prob, backend, dtype):
    """prost.solve against oracle.Solver.solve: result string, iteration count, x / y / z / w, and every intermediate callback
    (iteration number and the iterates it was handed)"""
    tp, to = [], []
    res = prost.solve(prob, backend, solve_opts(c, prob, tp))
    prob.finalize()
    oo = solve_opts(c, prob, to)
    b = [backend[0], {k: v for k, v in backend[1].items() if k not in NOT_ORACLE}]
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, oo, dtype)
    s.initialize()
    sc = s.scalars()
    msg, iters = s.solve()
    ost = s.state()
    c["_steps0_oracle"] = (sc["tau"], sc["sigma"]); c["_steps0"] = c["_steps0_oracle"]
    if not all(np.isfinite(ost[v]).all() for v in "xyzw") or not all(np.isfinite(t[1]).all() and np.isfinite(t[2]).all() for t in to):
        return "skip", res.get("path")                     # a random composition that diverges
    if msg != res["result"] or int(iters) != int(res["iters"]):
        return "solve: %r after %r iterations vs %r after %r" % (res["result"], res["iters"], msg, iters), res.get("path")
    if [t[0] for t in tp] != [t[0] for t in to]:
        return "solve: callbacks at %s vs %s" % ([t[0] for t in tp], [t[0] for t in to]), res.get("path")
    for (i, x, y), (_, ox, oy) in zip(tp, to):
        if not (np.array_equal(np.ravel(x), np.ravel(ox)) and np.array_equal(np.ravel(y), np.ravel(oy))):
            return "solve: iterates handed to the callback at iteration %d differ (x %.3g, y %.3g)" % (i, float(np.abs(np.ravel(x) - np.ravel(ox)).max()), float(np.abs(np.ravel(y) - np.ravel(oy)).max())), res.get("path")
    for v in "xyzw":
        if not np.array_equal(np.ravel(np.asarray(res[v])), ost[v]):
            return "solve: final %s (max |d| %.3g)" % (v, float(np.abs(np.ravel(np.asarray(res[v])) - ost[v]).max())), res.get("path")
    return None, res.get("path")


def product(prob, backend, opts, c):
    s = prost.Solver(prob, backend, opts)
    st0 = s.state(vectors=False)
    c["_steps0"] = (st0["tau"], st0["sigma"])
    if c["checked"]:                                     # the loop of prost.solve, in two calls (a speculative launch is pending in between)
        k = c["iters"] // 2
        if k:
            s.iterate(k, checked=True)
        s.iterate(c["iters"] - k, checked=True)
    else:
        s.iterate(c["iters"])
    st = s.state()
    s.destroy()
    return st


def reference(prob, backend, opts, c, dtype):
    prob.finalize()
    b = [backend[0], {k: v for k, v in backend[1].items() if k not in NOT_ORACLE}]
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, opts, dtype)
    s.initialize()
    sc = s.scalars()
    c["_steps0_oracle"] = (sc["tau"], sc["sigma"])
    s.iterate(c["iters"])
    st = s.state()
    st.update(s.scalars())
    return st


def close(a, b, dtype, tol=None):
    tol = tol or (1e-4 if dtype == np.float32 else 1e-10)
    for v in "xyzw":
        if float(np.abs(a[v] - b[v]).max()) > tol * max(1.0, float(np.abs(b[v]).max())):
            return False
    return True


def differs(a, b):
    for v in "xyzw":
        if not np.array_equal(a[v], b[v]):
            return "%s (max |d| %.3g, %d elements)" % (v, float(np.abs(a[v] - b[v]).max()), int((a[v] != b[v]).sum()))
    for v in ("tau", "sigma", "theta"):
        if a[v] != b[v]:
            return "%s %r vs %r" % (v, a[v], b[v])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=600.0)
    ap.add_argument("--mode", choices=["fused", "generic", "large", "sharded"], default="fused")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    prost.set_gpu(0)
    if args.mode == "large":
        return run_large(args, rng)
    if args.mode == "sharded":
        return run_sharded(args, rng)
    t0, done, fails, ties, setups, skipped, inexact, diverged, sensitive, solves, paths = time.time(), 0, 0, 0, 0, 0, 0, 0, 0, 0, {}
    admm_exact = 0
    for i in range(args.cases):
        if time.time() - t0 > args.budget_s:
            break
        c = draw(rng) if args.mode == "fused" else draw_generic(rng)
        prost.set_precision(c["precision"])
        dtype = np.float32 if c["precision"] == "single" else np.float64
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        if c.get("backend") == "admm":
            b = prost.backend.admm(rho0=float(np.random.default_rng(c["seed"]).choice([0.5, 1.0, 4.0])), residual_iter=c["residual_iter"])
        else:
            b = prost.backend.pdhg(stepsize=c["step"], residual_iter=c["residual_iter"], alg2_gamma=c["gamma"], scale_steps_operator=c["scale_steps"])
        # product-only switches (round 5), drawn per case: the operator inside the prox launches (0 never / 1 stencil operators / 2 any CSR or
        # gradient operator), the step rule on the device or on the host, ADMM's two-launch CG rounds
        r5 = np.random.default_rng(c["seed"] + 505)
        if c.get("backend") == "admm":
            b[1]["pixel_rounds"] = bool(r5.random() < 0.7)
        else:
            b[1]["allow_op_fusion"] = int(r5.choice([0, 0, 2, 2]))
            b[1]["residual_sums_in_prox"] = int(r5.choice([0, 1, 2, 2]))
            b[1]["allow_device_rules"] = bool(r5.random() < 0.7)
            # round 6: up to four iterations per launch in the EXACT class (kernels_fused_iterk.hip with the exact forms; the shapes it does not
            # take keep their pair / single launches): every partition of the iterations must give the oracle's iterates bit for bit
            b[1]["group_max"] = int(r5.choice([1, 1, 2, 3, 4]))
        product_only = {k: b[1][k] for k in ("pixel_rounds", "allow_op_fusion", "residual_sums_in_prox", "allow_device_rules", "group_max") if k in b[1]}
        c["product_only"] = product_only
        try:
            prob = build(c) if args.mode == "fused" else build_generic(c)
            if "solve" in c and (c.get("long_rows") or "transcendental" in c.get("_desc", ())):
                del c["solve"]                                            # (tolerance classes: compared after a fixed number of iterations below)
            if "solve" in c:
                if c["step"] in ("goldstein", "boyd") or c["scale_steps"]:
                    c["step"] = "alg2" if c["seed"] % 2 else "alg1"       # (exact comparison of a whole solve: no residual-driven steps, no norm estimate)
                    c["scale_steps"] = False
                    b = prost.backend.pdhg(stepsize=c["step"], residual_iter=c["residual_iter"], alg2_gamma=c["gamma"], scale_steps_operator=False)
                    b[1].update(product_only)
                d, path = compare_solve(c, prob, b, dtype)
                paths[path] = paths.get(path, 0) + 1
                if d == "skip":
                    skipped += 1
                    continue
                done += 1; solves += 1
                if d:
                    fails += 1
                    print("FAIL %s: %s" % ({k: v for k, v in c.items() if not k.startswith("_")}, d), flush=True)
                continue
            st = product(prob, b, o, c)
            paths[st["path"]] = paths.get(st["path"], 0) + 1
            ost = reference(prob, b, o, c, dtype)
            if not all(np.isfinite(ost[v]).all() for v in "xyzw"):      # a random composition that diverges: nothing to compare
                skipped += 1
                continue
            if c.get("backend") == "admm":
                # CG step lengths come from reductions: tolerance (tests/test_gpu_solver.py: the a5 bar), and equal CG iteration counts
                # (continuous operations only, see _function; a CG solve that stops one round earlier or later on a tie of its
                # tolerance test is counted, and compared with ten times the tolerance)
                # (double: the CG solves stop at a tolerance of their own -- two runs whose reductions round differently agree to about
                # that tolerance times the conditioning of the random operator, not to 1e-9 as on the TV-L1 problem of the tests)
                tol = 2e-4 if dtype == np.float32 else 2e-6
                ok = close(st, ost, dtype, tol)
                admm_exact += 1 if (all(np.array_equal(st[v], ost[v]) for v in "xyzw") and st["cg_iterations"] == ost["cg_iterations"]) else 0
                if not ok and st["cg_iterations"] != ost["cg_iterations"] and close(st, ost, dtype, 10 * tol):
                    ties += 1
                    ok = True
                if not ok:
                    # how far the ORACLE itself moves when rho0 changes in its last place: some random compositions (indicator
                    # functions on both sides, CG stopped by its iteration limit) amplify a rounding difference by many orders of
                    # magnitude within a few iterations -- the product may be as far from the oracle as the oracle is from itself
                    b2 = [b[0], dict(b[1], rho0=float(dtype(b[1]["rho0"]) * (dtype(1) + np.finfo(dtype).eps)))]
                    pst = reference(prob, b2, o, c, dtype)
                    ok = all(float(np.abs(st[v] - ost[v]).max()) <= tol * max(1.0, float(np.abs(ost[v]).max())) + 1e3 * float(np.abs(pst[v] - ost[v]).max()) for v in "xyzw")
                    sensitive += 1 if ok else 0
                d = None if ok else "admm: %s; cg iterations %r vs %r" % (differs(st, ost), st["cg_iterations"], ost["cg_iterations"])
                done += 1
                if d:
                    fails += 1
                    print("FAIL %s: %s%s" % ({k: v for k, v in c.items() if not k.startswith("_")}, d, "  [" + " ; ".join(c["_desc"]) + "]" if c.get("_desc") else ""), flush=True)
                continue
            d = differs(st, ost)
            if d and (c.get("long_rows") or "transcendental" in c.get("_desc", ())):
                # (sums in another order: within the tolerance, or -- a residual threshold or a discontinuous operation downstream -- anywhere)
                inexact += 1
                diverged += 0 if close(st, ost, dtype) else 1
                d = None
            if d and c["_steps0"] != c["_steps0_oracle"]:
                t0p, t0o = c["_steps0"][0], c["_steps0_oracle"][0]
                # Round 5: the norms of the power iteration are order-independent sums on both sides (reduce.hpp dd_t / the oracle's ExactSum),
                # so the estimate -- and with it the rescaled initial steps -- must be the SAME bits; rounds 3-4 allowed 128 ulp here and
                # counted such cases as "rescaled initial steps".  (Operators with long sparse rows, whose products are summed by
                # cooperating lanes, never get here: they are compared with a tolerance above.)
                d = "initial tau %r vs %r (operator-norm estimate differs); then %s" % (t0p, t0o, d)
            if d and c["step"] in ("goldstein", "boyd") and args.mode == "generic":
                # no second product path to ask: a residual threshold that decided differently shows in the step sizes
                tie = st["tau"] != ost["tau"] or st["sigma"] != ost["sigma"]
                if not tie:
                    # (boyd can divide and later multiply by the same factor: equal steps at the end, different ones on the way) -- the same
                    # composition under alg1, whose steps do not look at the residuals, must then be exact
                    b1 = prost.backend.pdhg(stepsize="alg1", residual_iter=c["residual_iter"], alg2_gamma=c["gamma"], scale_steps_operator=c["scale_steps"])
                    p1 = build_generic(c)
                    tie = differs(product(p1, b1, o, c), reference(p1, b1, o, c, dtype)) is None
                if tie:
                    ties += 1
                    d = None
            elif d and c["step"] in ("goldstein", "boyd"):
                bg = prost.backend.pdhg(stepsize=c["step"], residual_iter=c["residual_iter"], alg2_gamma=c["gamma"], scale_steps_operator=c["scale_steps"])
                bg[1]["allow_fused"] = False
                if differs(st, product(build(c) if args.mode == "fused" else build_generic(c), bg, o, c)) is None:
                    ties += 1
                    print("tie  %s: %s (fused == generic product paths; a residual threshold decided differently in T)" % (c, d), flush=True)
                    d = None
        except Exception as e:                                      # noqa: BLE001 -- a fuzz harness reports whatever went wrong
            d = "exception: %s" % e
        done += 1
        if d:
            fails += 1
            print("FAIL %s: %s%s" % ({k: v for k, v in c.items() if not k.startswith("_")}, d, "  [" + " ; ".join(c["_desc"]) + "]" if c.get("_desc") else ""), flush=True)
    prost.set_precision("double")
    print("fuzz_parity: %d cases (%d of them complete solves with callbacks) in %.0f s, %d failures, %d threshold ties, %d with rescaled initial steps (setup; %d of them not within the tolerance afterwards), %d divergent compositions skipped, %d with long sparse rows or a transcendental projection compared with a tolerance, %d ADMM cases as sensitive in the oracle itself, %d ADMM cases bit-identical to the oracle (iterates and CG iteration counts), paths %s"
          % (done, solves, time.time() - t0, fails, ties, setups, diverged, skipped, inexact, sensitive, admm_exact, paths))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
