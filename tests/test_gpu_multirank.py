"""The native N > 1 logic with TWO REAL RANKS on one GPU.

RCCL refuses two ranks on the same device, so the ranks use the host-callback transport of the communicator
(prost_hip_comm_create_host / prost_comm_init_host: the all-reduce is enqueued as D2H copy + host function + H2D copy
and ordered by the stream exactly like the RCCL call) with gloo between the processes.  Everything above the transport
is the production path: global sizes in eps_primal / eps_dual, the side-stream all-reduce with its event hand-over
(alg1 / alg2), the in-stream all-reduce of the residual-driven rules (boyd / goldstein), identical step sizes and
stopping decisions on every rank.  The checks are those tests/test_distributed_cpu.py makes on the oracle -- here the
product is compared WITH the oracle run under the same all-reduce.

Worker processes come from the fork server conftest.py starts before the GPU is touched (tests/multirank_workers.py).
"""
import json
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

import multirank_workers as workers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("step,precision", [("alg2", "single"), ("boyd", "single"), ("alg1", "double"), ("goldstein", "double")])
def test_two_ranks_on_one_gpu_match_the_oracle_under_the_same_allreduce(hip, step, precision):
    ctx = mp.get_context("forkserver")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=workers.pdhg_rank, args=(r, 2, port, step, precision, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=60)
    for d in res:
        assert "error" not in d, d.get("error")
    a, b = res
    assert a["path"] == b["path"] == "pdhg:fused-grad2d"
    assert a["xsum"] != b["xsum"]                                           # different problems (seeds 42, 43)
    for d in res:
        assert all(d["same"].values()), (d["rank"], d["same"])             # iterates == oracle, bit for bit
        assert d["mid_iteration"] == 37
        assert d["n_calls"] >= 1 + 30                                       # global sizes + one all-reduce per residual iteration
        for k in ("tau", "sigma", "theta"):
            assert d["scal"][k] == d["oscal"][k], (k, d["scal"][k], d["oscal"][k])
        for k in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual"):
            assert np.isclose(d["scal"][k], d["oscal"][k], rtol=1e-5), (k, d["scal"][k], d["oscal"][k])
    # every rank sees the same global scalars
    for k in a["scal"]:
        assert a["scal"][k] == b["scal"][k], k
    # eps uses the GLOBAL sizes: sqrt(sum m) * tol_abs + tol_rel * global norm (backend.hpp:71-74)
    m = 2 * a["nrows"]
    assert np.isclose(a["scal"]["eps_primal"], np.sqrt(m) * 1e-4 + 1e-4 * a["scal"]["primal_var_norm"], rtol=1e-5)
    # the full solve stops on the global criterion: same iteration, same verdict on both ranks
    assert a["solve_iters"] == b["solve_iters"] and a["solve_iters"] < 4000 and a["solve_result"] == b["solve_result"] == "Converged."


@pytest.mark.parametrize("world,precision,L,nx,ny,halo", [(2, "single", 1, 64, 252, 6), (3, "double", 1, 96, 64, 8), (2, "single", 3, 48, 128, 7), (3, "single", 2, 66, 64, 5)])
def test_column_slabs_on_real_ranks_exchange_halos_between_processes(hip, world, precision, L, nx, ny, halo):
    """SURVEY 8f.4 with REAL ranks: every rank owns one column slab in its own process and HIP context (all on GPU 0), the
    native loop solver_iterate_sharded exchanges the halo columns through prost_hip_comm_send / recv on the host-callback
    transport (D2H, gloo isend / irecv of the whole group, H2D -- where RCCL would run ncclSend / ncclRecv) and the residual
    sums through its all-reduce.  Owned columns of every rank == the oracle's iterates of the WHOLE image, bit for bit; the
    all-reduced residuals == the oracle's residuals of the whole image."""
    ctx = mp.get_context("forkserver")
    out = ctx.Queue()
    port = _free_port()
    iters = (13, 45)
    procs = [ctx.Process(target=workers.sharded_rank, args=(r, world, port, precision, L, nx, ny, halo, iters, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=60)
    for d in res:
        assert "error" not in d, d.get("error")
    exchanges = (iters[1] - 1) // (halo - 2)           # one refresh every halo - 2 iterations, none before the first
    for d in res:
        assert all(d["same"].values()), (d["rank"], d["same"])
        assert d["iteration"] == iters[1] and d["nranks"] == world and d["transport"] == "host"
        assert d["path"].startswith("pdhg:fused-grad2d"), d["path"]
        assert d["exchanges"] == exchanges, (d["exchanges"], exchanges)
        sides = (1 if d["rank"] > 0 else 0) + (1 if d["rank"] < world - 1 else 0)
        assert all(b == sides * 3 * L * halo * ny * d["itemsize"] for b in d["bytes_sent"]), d["bytes_sent"]
        # every rank reports the residuals of the WHOLE image (owned columns only, summed over the ranks)
        assert np.isclose(d["primal_res"], d["o_primal_res"], rtol=2e-5) and np.isclose(d["dual_res"], d["o_dual_res"], rtol=2e-5)
    assert len({(d["primal_res"], d["dual_res"]) for d in res}) == 1


def _run_bench(cmd, env):
    ctx = mp.get_context("forkserver")
    out = ctx.Queue()
    p = ctx.Process(target=workers.run_command, args=(cmd, env, ROOT, out))
    p.start()
    rc, stdout, stderr = out.get(timeout=600)
    p.join(timeout=60)
    return rc, stdout, stderr


def test_bench_starts_its_own_ranks_without_a_launcher(hip):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py itself starts the two ranks (a child
    torch.distributed.run job created before anything touches the GPU) and prints the rank-0 line with n_gpus == 2;
    a launcher whose WORLD_SIZE disagrees with --gpus is refused instead of silently reporting another n_gpus."""
    env = {"PROST_BENCH_TRANSPORT": "host", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        assert k not in os.environ
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--size", "1024", "--prelude-iters", "100", "--no-cpu-baseline"]
    rc, stdout, stderr = _run_bench(cmd, env)
    assert rc == 0, stderr
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["problems"] == 2 and d["config"]["comm_nranks"] == 2 and d["config"]["rccl_nranks"] is None
    assert d["value"] > 0 and d["iterates_finite"]
    # WORLD_SIZE = 1 from a launcher, --gpus 2 on the command line: refused
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--size", "256", "--prelude-iters", "0", "--no-cpu-baseline"]
    rc, stdout, stderr = _run_bench(cmd, env)
    assert rc != 0 and "WORLD_SIZE = 1" in stderr, (rc, stderr[-500:])
    assert not [l for l in stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_under_torchrun_on_one_gpu(hip):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on GPU 0
    over the host-callback transport (PROST_BENCH_TRANSPORT=host): rank-0 JSON assembly, barrier and max-over-ranks
    timing, weak-scaling bookkeeping -- executed once before a real 8-GPU run"""
    ctx = mp.get_context("forkserver")
    out = ctx.Queue()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--size", "1024",
           "--prelude-iters", "100", "--no-cpu-baseline"]
    p = ctx.Process(target=workers.run_command, args=(cmd, {"PROST_BENCH_TRANSPORT": "host", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, ROOT, out))
    p.start()
    rc, stdout, stderr = out.get(timeout=600)
    p.join(timeout=60)
    assert rc == 0, stderr
    line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["config"]["problems"] == 2
    assert d["config"]["residual_allreduce"] == "host-callback (gloo)"
    assert d["value"] > 0 and d["iterates_finite"] and d["roofline"]["launches_timed"] >= 3
    assert "cpu_baseline" not in d


def test_bench_eight_ranks_on_one_gpu(hip):
    """BASELINE config 5 as the driver will launch it on an 8-GPU node -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` --
    with the eight ranks sharing the one GPU there is (PROST_BENCH_TRANSPORT=host: gloo under the native communicator's host-callback
    transport): rendezvous of eight processes, eight independent problems (seeds 42 .. 49), global sizes in the stopping criterion,
    barriers and MAX-reduced timing, ONE rank-0 JSON line with n_gpus = 8 and the aggregate rate, exit code 0.  Also without a launcher
    (bench.py starts its eight ranks itself).  A functional run of the N = 8 code path, not a scaling figure."""
    env = {"PROST_BENCH_TRANSPORT": "host", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "1"}
    base = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--size", "1024", "--prelude-iters", "40", "--no-cpu-baseline"]
    for launcher in (True, False):
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
               if launcher else [sys.executable]) + [os.path.join(ROOT, "bench.py")] + base
        rc, stdout, stderr = _run_bench(cmd, env)
        assert rc == 0, stderr[-3000:]
        lines = [l for l in stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
        assert d["config"]["problems"] == 8 and d["config"]["comm_nranks"] == 8 and d["config"]["rccl_nranks"] is None
        assert d["config"]["residual_allreduce"] == "host-callback (gloo)" and d["config"]["rendezvous"].startswith("gloo")
        assert d["value"] > 0 and d["iterates_finite"] and abs(d["value"] - 8 * 20 / (d["ms_per_step"] * 20e-3)) <= 1e-6 * d["value"]
        assert d["metric"] == "PDHG iters/sec, ROF-TV 1024^2 fp32" and "cpu_baseline" not in d


@pytest.mark.parametrize("inject", [None, "injected by the test"])
def test_bench_native_rccl_communicator_and_its_fallback(hip, inject):
    """the RCCL leg of bench.py on the one GPU there is (PROST_BENCH_FORCE_DIST=1): the rendezvous runs over gloo on CPU tensors, so
    the solver's own communicator is the ONLY RCCL communicator of the rank (RCCL's own log shows exactly one initialisation) and
    the residual all-reduce runs through it.  When that communicator cannot be set up the run goes on over the host-callback
    transport, loudly -- and does not count: "value" and "rccl_nranks" are null in the line, the process exits non-zero."""
    env = {"PROST_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_PORT": str(_free_port()), "NCCL_DEBUG": "INFO"}
    if inject:
        env["PROST_BENCH_INJECT_RCCL_FAILURE"] = inject
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--size", "1024", "--prelude-iters", "100", "--no-cpu-baseline"]
    rc, stdout, stderr = _run_bench(cmd, env)
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (rc, stderr[-2000:])
    d = json.loads(lines[-1])
    inits = [l for l in (stdout + stderr).splitlines() if "Init COMPLETE" in l]
    assert d["n_gpus"] == 1 and d["iterates_finite"] and d["config"]["comm_nranks"] == 1 and d["config"]["rendezvous"].startswith("gloo")
    if inject:
        assert rc == 3, (rc, stderr[-2000:])
        assert d["value"] is None and d["value_without_rccl"] > 0 and d["config"]["rccl_nranks"] is None and "RCCL" in d["error"]
        assert d["config"]["residual_allreduce"].startswith("host-callback (gloo) after")
        assert "native RCCL communicator failed (injected by the test)" in stderr
        assert len(inits) == 0, inits                      # nothing created an RCCL communicator behind the solver's back
    else:
        assert rc == 0, stderr[-2000:]
        assert d["value"] > 0 and d["config"]["residual_allreduce"] == "rccl" and d["config"]["rccl_nranks"] == 1
        assert len(inits) == 1, inits                      # ONE communicator per rank: the solver's
