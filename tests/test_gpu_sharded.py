"""ONE image sharded by column slabs (prost_amd.distributed.ColumnShardedSolver): every slab
runs the unmodified kernels on its slab + halo columns and refreshes the halos every halo - 2 iterations.
On a 1-GPU box the slabs live in one process (in-process halo copies); the RCCL transport is exercised
with a rank that is its own neighbour (world 1) and must move exactly the same bytes."""
import numpy as np
import pytest

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


def _maker(f_full, ny):
    def make(lo, hi):
        prob, _, _, _ = synthetic.rof_problem(hi - lo, ny, f=f_full[lo * ny: hi * ny])
        return prob
    return make


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("world,halo,nx,ny", [(3, 8, 96, 64), (2, 5, 41, 252), (4, 12, 130, 500), (3, 6, 60, 249)])
def test_slabs_reproduce_the_single_image_iterates(prec, dtype, world, halo, nx, ny):
    prost.set_precision(prec)
    f = np.asarray(synthetic.rof_image(nx, ny, 1, 9)).ravel()
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    opts = prost.options(**OPTS)
    for iters in (7, 45):
        full = prost.Solver(_maker(f, ny)(0, nx), [backend[0], dict(backend[1], scale_steps_operator=False)], opts)
        full.iterate(iters)
        fs = full.state(); full.destroy()
        slabs = [distributed.ColumnShardedSolver(_maker(f, ny), nx, ny, backend, opts, r, world, halo, transport="local") for r in range(world)]
        for s in slabs:
            s.transport = slabs
        distributed.iterate_group(slabs, iters)
        n = nx * ny
        res2 = np.zeros(2)
        for s in slabs:
            st = s.owned_state()
            a, b = s.c0 * ny, s.c1 * ny
            assert np.array_equal(st["x"], fs["x"][a:b]), (s.rank, "x")
            assert np.array_equal(st["y1"], fs["y"][a:b]) and np.array_equal(st["y2"], fs["y"][n + a:n + b]), (s.rank, "y")
            assert st["iteration"] == iters
            res2 += [st["primal_res"] ** 2, st["dual_res"] ** 2]
            s.destroy()
        # the residual sums of the slabs (owned columns only) add up to the single-image residuals
        assert np.allclose(np.sqrt(res2), [fs["primal_res"], fs["dual_res"]], rtol=2e-5), (np.sqrt(res2), fs["primal_res"], fs["dual_res"])


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
