"""Multi-GPU batches: independent problems sharded one (or more) per rank, with a GLOBAL stopping
criterion -- the four residual sums (sum diff_p^2, sum z^2, sum diff_d^2, sum w^2) are all-reduced
so every rank takes the same step-size / stopping decisions (the reference itself is
single-GPU).  No data-path collective: iterates never leave their GPU.
"""
import numpy as np


def shard(num_problems, rank, world):
    """problem ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(num_problems, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def problem_seed(problem_id, base_seed=42):
    """BASELINE config 5: seeds 42..49 for the eight 4096^2 problems"""
    return base_seed + problem_id


def init_native_comm(dist, device):
    """Create the RCCL communicator owned by the native solver (prost_hip_comm_create): rank 0
    makes the ncclUniqueId, torch.distributed broadcasts its 128 bytes."""
    import torch

    from . import _capi
    ident = torch.zeros(128, dtype=torch.float64, device=device)
    if dist.get_rank() == 0:
        ident.copy_(torch.from_numpy(_capi.comm_unique_id()))
    dist.broadcast(ident, src=0)
    _capi.comm_init(ident.cpu().numpy(), dist.get_rank(), dist.get_world_size())


def allreduce_hook(dist):
    """host-side hook with the same contract (sum 4 doubles in place) for CPU/gloo runs"""
    import torch

    def hook(v4):
        t = torch.from_numpy(np.array(v4, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()
    return hook


# ------------------------------------------------------------------------------------------
# ONE image sharded over ranks by column slabs
# ------------------------------------------------------------------------------------------
def column_slab(nx, rank, world, halo):
    """Owned image columns [c0, c1) of `rank` and its halo widths (left, right): contiguous slabs whose
    sizes differ by at most one; no halo at the image border."""
    base, extra = divmod(nx, world)
    c0 = rank * base + min(rank, extra)
    c1 = c0 + base + (1 if rank < extra else 0)
    return c0, c1, (halo if rank > 0 else 0), (halo if rank < world - 1 else 0)


class ColumnShardedSolver:
    """PDHG on ONE gradient2d image split into column slabs, one slab (+ `halo` columns per inner side)
    per solver.  Every slab runs the unmodified single-image kernels on its extended slab; the wrong
    boundary condition at an artificial edge contaminates one more column per iteration, so after at
    most halo - 2 iterations the halo columns of x and y are refreshed from the neighbours' owned
    columns (communication-avoiding: one exchange of 3 * halo * ny values per side every halo - 2
    iterations instead of one per operator application).  Owned columns are then bit-identical to the
    single-GPU iterates.  Residual sums count owned columns only and are all-reduced through the native
    communicator when there is one, so step sizes and stopping agree on all ranks.

    make_problem(col_lo, col_hi) must build the problem restricted to image columns [col_lo, col_hi)
    (an nx' = col_hi - col_lo wide image); the backend must not rescale the step sizes from a LOCAL
    operator norm, so `scale_steps_operator` is forced off (for gradient operators the reference's
    rescaling branch |norm - 1| > 0.1 never fires anyway).

    transport: "comm" / "rccl" (one slab per rank through the native communicator: prost.comm_init = RCCL over xGMI, or
    prost.comm_init_host with a p2p function = host transport, several ranks on one GPU) or a list of ColumnShardedSolver
    objects of the same process (several slabs on one GPU)."""

    def __init__(self, make_problem, nx, ny, backend, opts, rank, world, halo=8, transport="comm"):
        from . import _capi
        if halo < 3:
            raise ValueError("halo must be at least 3 columns")
        self.nx, self.ny, self.rank, self.world, self.halo = nx, ny, rank, world, halo
        self.c0, self.c1, self.hl, self.hr = column_slab(nx, rank, world, halo)
        if self.c1 - self.c0 < halo:
            raise ValueError("slab of %d columns is narrower than the halo" % (self.c1 - self.c0))
        lo, hi = self.c0 - self.hl, self.c1 + self.hr
        self.nl = hi - lo
        prob = make_problem(lo, hi)
        backend = [backend[0], dict(backend[1], scale_steps_operator=False)]
        self.solver = _capi.Solver(prob, backend, opts, owned_columns=(self.hl, self.hl + (self.c1 - self.c0), self.nl))
        self.transport = "comm" if transport == "rccl" else transport
        self.since_exchange = 0

    def exchange(self, peers=None):
        h, ny = self.halo, self.ny
        if self.transport == "comm":
            if self.hl or self.hr:
                self.solver.halo_exchange(ny, h, self.hl, self.hr, self.rank - 1 if self.hl else -1, self.rank + 1 if self.hr else -1)
        else:
            peers = peers if peers is not None else self.transport
            if self.hl:                       # left halo <- left neighbour's last owned columns
                src = peers[self.rank - 1]
                self.solver.copy_columns_from(0, src.solver, src.nl - src.hr - h, h, ny)
            if self.hr:                       # right halo <- right neighbour's first owned columns
                src = peers[self.rank + 1]
                self.solver.copy_columns_from(self.nl - self.hr, src.solver, src.hl, h, ny)
        self.since_exchange = 0

    def steps_until_exchange(self):
        return (self.halo - 2) - self.since_exchange

    def iterate_local(self, k):
        self.solver.iterate(k)
        self.since_exchange += k

    def iterate(self, iters):
        """communicator transport: runs `iters` iterations with the exchanges in between (collective: every rank
        calls it with the same count)"""
        if self.transport != "comm":
            raise RuntimeError("in-process slabs are driven by iterate_group()")
        # the exchange / iterate loop runs inside the native solver (solver_iterate_sharded): no Python between exchanges
        self.since_exchange = self.solver.iterate_sharded(iters, self.ny, self.halo, self.hl, self.hr, self.rank - 1 if self.hl else -1,
                                                          self.rank + 1 if self.hr else -1, self.since_exchange)

    def owned_state(self):
        st = self.solver.state()
        P, ny = self.nl * self.ny, self.ny
        a, b = self.hl * ny, (self.hl + self.c1 - self.c0) * ny
        x, y = np.asarray(st["x"]), np.asarray(st["y"])
        L = x.size // P
        out = {"x": np.concatenate([x[l * P + a:l * P + b] for l in range(L)]),
               "y1": np.concatenate([y[l * P + a:l * P + b] for l in range(L)]),
               "y2": np.concatenate([y[(L + l) * P + a:(L + l) * P + b] for l in range(L)]),
               "iteration": st["iteration"], "primal_res": st["primal_res"], "dual_res": st["dual_res"]}
        return out

    def destroy(self):
        self.solver.destroy()


def iterate_group(slabs, iters):
    """several slabs in one process: lock-step iterations with in-process halo copies"""
    done = 0
    while done < iters:
        if slabs[0].steps_until_exchange() <= 0:
            for s in slabs:
                s.exchange(slabs)
        k = min(iters - done, slabs[0].steps_until_exchange())
        for s in slabs:
            s.iterate_local(k)
        done += k
