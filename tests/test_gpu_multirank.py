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
    assert d["value"] > 0 and d["iterates_finite"] and d["roofline"]["launches_timed"] >= 5
    assert "cpu_baseline" not in d
