#!/usr/bin/env python3
"""ONE ROF image sharded by column slabs over the GPUs of a node:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_rof.py --size 8192 --steps 400
Each rank owns nx / N columns (+ `--halo` halo columns per inner side), exchanges 3 * halo * ny values with each
neighbour over RCCL every halo - 2 iterations and all-reduces the four residual sums at residual iterations.
Prints one JSON line (aggregate iterations/second of the ONE image = strong scaling)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--halo", type=int, default=12)
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch

    import prost_amd as prost
    from prost_amd import distributed, synthetic

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    prost.set_gpu(local_rank)
    prost.set_precision("single")
    if world > 1:
        distributed.init_native_comm(dist, torch.device("cuda", local_rank))
    n = args.size

    def make(lo, hi):
        # every rank generates only its own columns of the counter-hashed image (same pixels as the full image)
        f = np.asarray(synthetic.rof_image(n, n, 1, 42)).ravel()[lo * n: hi * n] if world == 1 else _columns(n, lo, hi)
        prob, _, _, _ = synthetic.rof_problem(hi - lo, n, f=f)
        return prob

    def _columns(nn, lo, hi):
        idx = np.arange(lo * nn, hi * nn, dtype=np.uint64)
        return synthetic.rof_image_at(nn, nn, idx, 42) if hasattr(synthetic, "rof_image_at") else np.asarray(synthetic.rof_image(nn, nn, 1, 42)).ravel()[lo * nn: hi * nn]

    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    opts = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    s = distributed.ColumnShardedSolver(make, n, n, backend, opts, rank, world, args.halo, transport="rccl")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    s.iterate(args.warmup)
    barrier()
    t0 = time.perf_counter()
    s.iterate(args.steps)
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    st = s.owned_state()
    if rank == 0:
        print(json.dumps({"metric": "PDHG iters/sec, ONE ROF-TV %d^2 fp32 image sharded by columns" % n, "value": args.steps / float(el.item()),
                          "unit": "it/s", "n_gpus": world, "steps": args.steps, "scaling": "strong", "halo_columns": args.halo,
                          "exchange_every": args.halo - 2, "finite": bool(np.isfinite(st["x"]).all())}), flush=True)
    s.destroy()
    if world > 1:
        prost.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
