"""N > 1 path on CPU: world_size-2 gloo run of the batch-of-independent-problems scheme with the
global residual all-reduce (the oracle solver stands in for the GPU solver: same hook contract)."""
import os
import socket

import numpy as np
import pytest

from prost_amd import distributed


def test_shard_partitions_problems():
    for n in (1, 7, 8, 9):
        for world in (1, 2, 4, 8):
            ids = [i for r in range(world) for i in distributed.shard(n, r, world)]
            assert ids == list(range(n))
            sizes = [len(distributed.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert [distributed.problem_seed(i) for i in range(8)] == list(range(42, 50))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, step, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import oracle
    import prost_amd as prost
    from prost_amd import synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pid = distributed.shard(world, rank, world)[0]
        prob, u, q, f = synthetic.rof_problem(20, 16, seed=distributed.problem_seed(pid))
        prob.finalize()
        b = prost.backend.pdhg(stepsize=step, residual_iter=2, alg2_gamma=0.5)
        o = prost.options(max_iters=40, num_cback_calls=0, verbose=False)
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float64)
        s.set_allreduce(distributed.allreduce_hook(dist), world * prob.nrows, world * prob.ncols)
        s.initialize()
        s.iterate(40)
        sc = s.scalars()
        loc = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float64)
        loc.initialize(); loc.iterate(40)
        lsc = loc.scalars()
        out.put((rank, sc, lsc, float(s.state()["x"].sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("step,world", [("alg1", 2), ("boyd", 2), ("alg1", 8), ("boyd", 8)])
def test_global_residual_allreduce_gloo(step, world):
    """world 2 and world 8 (BASELINE config 5: eight independent problems, seeds 42 .. 49): every rank sees the same global scalars"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, step, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    g0, l0 = res[0][1], res[0][2]
    # every rank sees the same global scalars and takes the same step-size decisions
    for _, g, _, _ in res[1:]:
        for k in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual", "tau", "sigma"):
            assert g0[k] == g[k], k
    assert len({r[3] for r in res}) == world           # different problems (seeds 42 .. 42 + world - 1)
    if step == "alg1":
        # iterates do not depend on residuals: global^2 = sum of the local squares
        for k in ("primal_res", "dual_res", "dual_var_norm"):
            assert np.isclose(g0[k] ** 2, sum(r[2][k] ** 2 for r in res), rtol=1e-12), k
        # eps uses the GLOBAL sizes: sqrt(sum m) * tol_abs + tol_rel * global norm
        m = world * 2 * 20 * 16
        assert np.isclose(g0["eps_primal"], np.sqrt(m) * 1e-4 + 1e-4 * g0["primal_var_norm"], rtol=1e-12)


def test_bench_refuses_a_launcher_with_another_world_size():
    """`bench.py --gpus 8` under a launcher that started another number of ranks must not report n_gpus = 8: it exits before anything
    touches a GPU (this runs on the CPU box).  A --gpus below 1 is refused as well."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ws in ("1", "2", "4"):
        env = dict(os.environ, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode != 0 and ("WORLD_SIZE = %s" % ws) in p.stderr and "--gpus 8" in p.stderr, (ws, p.stderr[-400:])
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus must be at least 1" in p.stderr


def test_column_slabs_partition_the_image():
    """prost_amd.distributed.column_slab: contiguous slabs covering [0, nx) exactly, sizes differing by at
    most one, halo columns only on inner sides"""
    from prost_amd.distributed import column_slab
    for nx in (7, 96, 4096, 4099):
        for world in (1, 2, 3, 8):
            if nx < world:
                continue
            slabs = [column_slab(nx, r, world, 8) for r in range(world)]
            assert slabs[0][0] == 0 and slabs[-1][1] == nx
            assert all(slabs[r][1] == slabs[r + 1][0] for r in range(world - 1))
            sizes = [c1 - c0 for c0, c1, _, _ in slabs]
            assert max(sizes) - min(sizes) <= 1
            assert slabs[0][2] == 0 and slabs[-1][3] == 0
            assert all(s[2] == 8 for s in slabs[1:]) and all(s[3] == 8 for s in slabs[:-1])
