"""ONE image sharded by column slabs (prost_amd.distributed.ColumnShardedSolver): every slab
runs the unmodified kernels on its slab + halo columns and refreshes the halos every halo - 2 iterations.
The owned columns of every slab are compared with the CPU ORACLE's iterates of the unsharded image (and with the
unsharded product).  On a 1-GPU box the slabs live in one process (in-process halo copies); the RCCL transport and
the native exchange / iterate loop (solver_iterate_sharded) are exercised with a rank that is its own neighbour
(world 1) and must move exactly the same bytes."""
import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import distributed, synthetic

pytestmark = pytest.mark.gpu
PRECISIONS = [("single", np.float32), ("double", np.float64)]
OPTS = dict(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)


@pytest.fixture(autouse=True)
def _gpu(hip):
    prost.set_gpu(0)
    yield
    prost.set_precision("double")


def _maker(f_full, ny, L=1):
    nx = f_full.size // (ny * L)

    def make(lo, hi):
        f = np.concatenate([f_full[l * nx * ny + lo * ny: l * nx * ny + hi * ny] for l in range(L)])
        prob, _, _, _ = synthetic.rof_problem(hi - lo, ny, L, f=f)
        return prob
    return make


def _owned(v, planes, nx, ny, c0, c1):
    """columns [c0, c1) of every image plane of a full-image vector"""
    return np.concatenate([v[k * nx * ny + c0 * ny: k * nx * ny + c1 * ny] for k in planes])


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("world,halo,nx,ny,L", [(3, 8, 96, 64, 1), (2, 5, 41, 252, 1), (4, 12, 130, 500, 1), (3, 6, 60, 249, 1), (3, 7, 66, 128, 3), (2, 9, 48, 252, 4), (2, 6, 40, 64, 2)])
def test_slabs_reproduce_the_oracle_iterates_of_the_whole_image(prec, dtype, world, halo, nx, ny, L):
    """gray (single-kernel / pair path), two channels and RGB / 4 channels (multi-channel one-kernel path)"""
    prost.set_precision(prec)
    f = np.asarray(synthetic.rof_image(nx, ny, L, 9)).ravel()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    noscale = [backend[0], dict(backend[1], scale_steps_operator=False)]
    opts = prost.options(**OPTS)
    for iters in (7, 45):
        whole = _maker(f, ny, L)(0, nx)
        full = prost.Solver(whole, noscale, opts)
        full.iterate(iters)
        fs = full.state(); full.destroy()
        whole.finalize()
        orc = oracle.Solver(whole.data, whole.nrows, whole.ncols, noscale, opts, dtype)
        orc.initialize(); orc.iterate(iters)
        ost, osc = orc.state(), orc.scalars()
        for v in "xy":
            assert np.array_equal(fs[v], ost[v]), v                  # unsharded product == oracle (as everywhere else)
        slabs = [distributed.ColumnShardedSolver(_maker(f, ny, L), nx, ny, backend, opts, r, world, halo, transport="local") for r in range(world)]
        for s in slabs:
            s.transport = slabs
        distributed.iterate_group(slabs, iters)
        res2 = np.zeros(2)
        for s in slabs:
            st = s.owned_state()
            assert np.array_equal(st["x"], _owned(ost["x"], range(L), nx, ny, s.c0, s.c1)), (s.rank, "x vs oracle")
            assert np.array_equal(st["y1"], _owned(ost["y"], range(L), nx, ny, s.c0, s.c1)), (s.rank, "y (d/dx) vs oracle")
            assert np.array_equal(st["y2"], _owned(ost["y"], range(L, 2 * L), nx, ny, s.c0, s.c1)), (s.rank, "y (d/dy) vs oracle")
            assert st["iteration"] == iters
            res2 += [st["primal_res"] ** 2, st["dual_res"] ** 2]
            s.destroy()
        # the residual sums of the slabs (owned columns only) add up to the residuals of the whole image: product and oracle
        assert np.allclose(np.sqrt(res2), [fs["primal_res"], fs["dual_res"]], rtol=2e-5), (np.sqrt(res2), fs["primal_res"], fs["dual_res"])
        assert np.allclose(np.sqrt(res2), [osc["primal_res"], osc["dual_res"]], rtol=2e-5), (np.sqrt(res2), osc["primal_res"], osc["dual_res"])


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_rccl_transport_moves_the_same_columns(prec, dtype):
    """world-1 communicator, the rank is its own left and right neighbour: ncclSend / ncclRecv must leave
    exactly what the in-process column copies leave"""
    prost.set_precision(prec)
    nx, ny, halo = 40, 64, 6
    f = np.asarray(synthetic.rof_image(3 * nx, ny, 1, 4)).ravel()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    opts = prost.options(**OPTS)
    prost.comm_init(prost.comm_unique_id(), 0, 1)
    try:
        states = []
        for mode in ("rccl", "local"):
            s = distributed.ColumnShardedSolver(_maker(f, ny), 3 * nx, ny, backend, opts, 1, 3, halo, transport="rccl")   # middle slab: halos on both sides
            s.iterate_local(3)
            if mode == "rccl":
                s.solver.halo_exchange(ny, halo, s.hl, s.hr, 0, 0)
            else:
                s.solver.copy_columns_from(0, s.solver, s.hl, halo, ny)
                s.solver.copy_columns_from(s.nl - s.hr, s.solver, s.nl - s.hr - halo, halo, ny)
            s.iterate_local(4)
            states.append(s.solver.state()); s.destroy()
        for v in "xy":
            assert np.array_equal(states[0][v], states[1][v]), v
    finally:
        prost.comm_destroy()


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("L", [1, 3])
def test_native_sharded_loop_equals_the_python_driven_one(prec, dtype, L):
    """solver_iterate_sharded (exchange / iterate loop inside the native solver, RCCL transport; world-1 communicator with the
    rank as its own neighbour) leaves the state the Python-driven loop with in-process column copies leaves"""
    prost.set_precision(prec)
    nx, ny, halo, iters = 40, 64, 6, 23
    f = np.asarray(synthetic.rof_image(3 * nx, ny, L, 4)).ravel()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    opts = prost.options(**OPTS)
    prost.comm_init(prost.comm_unique_id(), 0, 1)
    try:
        states = []
        for mode in ("native", "python"):
            s = distributed.ColumnShardedSolver(_maker(f, ny, L), 3 * nx, ny, backend, opts, 1, 3, halo, transport="rccl")   # middle slab
            if mode == "native":
                s.since_exchange = s.solver.iterate_sharded(iters, ny, halo, s.hl, s.hr, 0, 0, 0)
                assert s.since_exchange == iters % (halo - 2) or s.since_exchange == halo - 2
            else:
                done = 0
                while done < iters:
                    if s.steps_until_exchange() <= 0:
                        s.solver.copy_columns_from(0, s.solver, s.hl, halo, ny)
                        s.solver.copy_columns_from(s.nl - s.hr, s.solver, s.nl - s.hr - halo, halo, ny)
                        s.since_exchange = 0
                    k = min(iters - done, s.steps_until_exchange())
                    s.iterate_local(k)
                    done += k
            states.append(s.solver.state()); s.destroy()
        for v in "xy":
            assert np.array_equal(states[0][v], states[1][v]), v
    finally:
        prost.comm_destroy()
