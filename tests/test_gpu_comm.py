"""RCCL path of the global stopping criterion on one GPU: a world-size-1 communicator exercises
prost_hip_comm_create / prost_hip_allreduce_sum_f64 and the solver's global-size bookkeeping
(the N > 1 logic itself is covered on CPU by tests/test_distributed_cpu.py)."""
import numpy as np
import pytest

import prost_amd as prost
from prost_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stepsize", ["boyd", "alg2"])
def test_world1_communicator_gives_identical_solve(hip, stepsize):
    """boyd: the all-reduce runs on the iteration stream (the step rule needs the global sums at once);
    alg2: on the side stream, overlapped with the next iterations, resolved when the state is read."""
    prost.set_gpu(0)
    prost.set_precision("single")
    try:
        prob, u, q, f = synthetic.rof_problem(64, 48)
        b = prost.backend.pdhg(stepsize=stepsize, residual_iter=2, alg2_gamma=0.5)
        o = prost.options(max_iters=60, num_cback_calls=0, verbose=False)
        s = prost.Solver(prob, b, o); s.iterate(60); ref = s.state(); s.destroy()
        ident = prost.comm_unique_id()
        assert ident.shape == (128,)
        assert prost.comm_info() == {"nranks": 0.0, "transport": "none"}
        prost.comm_init(ident, 0, 1)
        try:
            assert prost.comm_info() == {"nranks": 1.0, "transport": "rccl"}       # ncclCommCount
            s = prost.Solver(prob, b, o); s.iterate(60); st = s.state(); s.destroy()
        finally:
            prost.comm_destroy()
        for k in "xyzw":
            assert np.array_equal(st[k], ref[k]), k
        for k in ("primal_res", "dual_res", "eps_primal", "eps_dual", "tau", "sigma"):
            assert st[k] == ref[k], k
    finally:
        prost.set_precision("double")
