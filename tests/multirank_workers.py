"""Worker processes of the multi-rank GPU tests (tests/test_gpu_multirank.py).

They are started by the fork server that tests/conftest.py launches BEFORE anything initialises the GPU: a process
that has touched the GPU must not fork / exec others on the GPU boxes.  Every worker is one rank: its own process, its
own HIP context on GPU 0, gloo between the ranks, and the native solver's communicator on the host-callback transport
(prost_comm_init_host) -- the N > 1 logic of BackendPDHG (global sizes, side-stream all-reduce, buffer hand-over,
identical step-size decisions) with two real ranks on a one-GPU box.
"""
import os
import subprocess
import sys

import numpy as np


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch, dist


def pdhg_rank(rank, world, port, step, precision, out):
    """one rank of a batch of independent ROF problems (seeds 42 + rank) with the global residual all-reduce:
    product (host-callback communicator) next to the oracle (same gloo group), iterate / state and a full solve"""
    try:
        torch, dist = _init(rank, world, port)
        import oracle
        import prost_amd as prost
        from prost_amd import distributed, synthetic
        dtype = np.float32 if precision == "single" else np.float64
        prost.set_gpu(0)
        prost.set_precision(precision)
        calls = []

        def allreduce(a):                      # runs on a HIP runtime thread, in stream order
            calls.append(a.copy())
            dist.all_reduce(torch.from_numpy(a))

        prost.comm_init_host(allreduce, world)
        prob, u, q, f = synthetic.rof_problem(64, 48, seed=42 + rank)
        b = prost.backend.pdhg(stepsize=step, residual_iter=2, alg2_gamma=0.5)
        o = prost.options(max_iters=60, num_cback_calls=0, verbose=False)
        s = prost.Solver(prob, b, o)
        s.iterate(37)                          # odd count: ends between residual iterations
        mid = s.state(vectors=False)
        s.iterate(23)
        st = s.state()
        s.destroy()
        n_calls = len(calls)
        # the oracle under the same all-reduce (tests/test_distributed_cpu.py)
        prob.finalize()
        os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype)
        os_.set_allreduce(distributed.allreduce_hook(dist), world * prob.nrows, world * prob.ncols)
        os_.initialize()
        os_.iterate(60)
        ost, osc = os_.state(), os_.scalars()
        same = {v: bool(np.array_equal(st[v], ost[v])) for v in "xyzw"}
        # a complete solve that stops on the GLOBAL criterion: same iteration count on every rank
        o2 = prost.options(max_iters=4000, num_cback_calls=0, verbose=False, tol_rel_primal=2e-3, tol_rel_dual=2e-3, tol_abs_primal=2e-3, tol_abs_dual=2e-3)
        res = prost.solve(prob, b, o2)
        prost.comm_destroy()
        out.put(dict(rank=rank, same=same, scal={k: float(st[k]) for k in ("tau", "sigma", "theta", "primal_res", "dual_res", "eps_primal", "eps_dual", "primal_var_norm", "dual_var_norm")},
                     oscal={k: float(osc[k]) for k in ("tau", "sigma", "theta", "primal_res", "dual_res", "eps_primal", "eps_dual", "primal_var_norm", "dual_var_norm")},
                     mid_iteration=float(mid["iteration"]), n_calls=n_calls, path=st["path"], xsum=float(st["x"].sum()),
                     solve_iters=int(res["iters"]), solve_result=res["result"], nrows=prob.nrows, ncols=prob.ncols))
        dist.destroy_process_group()
    except Exception as e:                      # never leave the parent waiting
        import traceback
        out.put(dict(rank=rank, error=traceback.format_exc() + repr(e)))


def sharded_rank(rank, world, port, precision, L, nx, ny, halo, iters, out):
    """one rank of ONE image cut into column slabs: the native exchange / iterate loop (solver_iterate_sharded)
    with the halo columns travelling BETWEEN PROCESSES through the host-callback transport's point-to-point function (gloo
    isend / irecv), the residual sums through its all-reduce.  Owned columns are compared with the oracle's iterates of the
    whole image."""
    try:
        torch, dist = _init(rank, world, port)
        import oracle
        import prost_amd as prost
        from prost_amd import distributed, synthetic
        dtype = np.float32 if precision == "single" else np.float64
        prost.set_gpu(0)
        prost.set_precision(precision)
        moved = []

        gloo = prost.gloo_p2p(dist)

        def p2p(ops):                          # runs on a HIP runtime thread, in stream order
            moved.append(sum(a.size for s, peer, a in ops if s))
            gloo(ops)

        prost.comm_init_host(lambda a: dist.all_reduce(torch.from_numpy(a)), world, p2p=p2p)
        info = prost.comm_info()
        f_full = np.asarray(synthetic.rof_image(nx, ny, L, 9)).ravel()

        def make(lo, hi):
            f = np.concatenate([f_full[l * nx * ny + lo * ny: l * nx * ny + hi * ny] for l in range(L)])
            return synthetic.rof_problem(hi - lo, ny, L, f=f)[0]

        backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        opts = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        s = distributed.ColumnShardedSolver(make, nx, ny, backend, opts, rank, world, halo, transport="comm")
        s.iterate(iters[0])
        s.iterate(iters[1] - iters[0])         # the exchange phase carries over between calls
        st = s.owned_state()
        path = s.solver.state(vectors=False)["path"]
        c0, c1 = s.c0, s.c1
        s.destroy()
        prost.comm_destroy()
        whole = make(0, nx)
        whole.finalize()
        noscale = [backend[0], dict(backend[1], scale_steps_operator=False)]
        orc = oracle.Solver(whole.data, whole.nrows, whole.ncols, noscale, opts, dtype)
        orc.initialize()
        orc.iterate(iters[1])
        ost, osc = orc.state(), orc.scalars()

        def owned(v, planes):
            return np.concatenate([v[k * nx * ny + c0 * ny: k * nx * ny + c1 * ny] for k in planes])

        same = {"x": bool(np.array_equal(st["x"], owned(ost["x"], range(L)))),
                "y1": bool(np.array_equal(st["y1"], owned(ost["y"], range(L)))),
                "y2": bool(np.array_equal(st["y2"], owned(ost["y"], range(L, 2 * L))))}
        out.put(dict(rank=rank, same=same, iteration=int(st["iteration"]), primal_res=float(st["primal_res"]), dual_res=float(st["dual_res"]),
                     o_primal_res=float(osc["primal_res"]), o_dual_res=float(osc["dual_res"]), exchanges=len(moved), bytes_sent=moved,
                     nranks=int(info["nranks"]), transport=info["transport"], path=path, itemsize=np.dtype(dtype).itemsize))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        out.put(dict(rank=rank, error=traceback.format_exc() + repr(e)))


def run_command(cmd, env, cwd, out):
    """runs a launcher (torchrun) from a process that has never touched the GPU"""
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run(cmd, env=e, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    out.put((p.returncode, p.stdout[-60000:], p.stderr[-3000:]))      # (the bench line alone is ~10 000 characters)
