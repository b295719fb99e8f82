"""Synthetic benchmark inputs (no image files travel with the repo).

ROF phantom of SURVEY.md 8(d): piecewise-constant checkerboard + counter-hash noise so every
rank / CPU / GPU run produces identical data without an RNG library.
"""
import numpy as np


def hash32(seed, idx):
    """murmur3-style finaliser of (idx, seed) -> uint32 (vectorised)."""
    x = (np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x85EBCA6B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x.astype(np.uint64) * np.uint64(0xC2B2AE35) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def rof_image(nx, ny, L=1, seed=42, dtype=np.float64):
    """f in [0,1]^(ny x nx x L), returned flattened column-major (idx = y + x*ny + l*nx*ny),
    the layout block_gradient2d.cu:50 indexes."""
    bs = max(1, max(nx, ny) // 8)
    out = np.empty(nx * ny * L, dtype=dtype)
    yy = (np.arange(ny) // bs)[:, None]
    xx = (np.arange(nx) // bs)[None, :]
    base = (0.25 + 0.5 * ((yy + xx) % 2)).astype(np.float64)          # (ny, nx)
    for l in range(L):
        idx = np.arange(nx * ny, dtype=np.uint64) + np.uint64(l * nx * ny)
        u = hash32(seed, idx).astype(np.float64) / 4294967296.0
        clean = np.roll(base, l * bs // 2, axis=0).T.reshape(-1)       # column-major flatten
        out[l * nx * ny:(l + 1) * nx * ny] = np.clip(clean + 0.1 * (u - 0.5) * 2.0, 0.0, 1.0)
    return out


def rof_problem(nx, ny, L=1, lmb=10.0, seed=42, f=None, data_term="square"):
    """The problem of matlab/examples/example_rof_primaldual.m:15-28 on synthetic data (data_term='abs': TV-L1, example_tvl1.m)."""
    from . import block, function
    from .problem import variable, min_max_problem
    if f is None:
        f = rof_image(nx, ny, L, seed)
    u = variable(nx * ny * L)
    q = variable(2 * nx * ny * L)
    prob = min_max_problem([u], [q])
    prob.add_function(u, function.sum_1d(data_term, 1, f, lmb))
    prob.add_function(q, function.sum_norm2(2 * L, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, block.gradient2d(nx, ny, L))
    return prob, u, q, f


def tv3d_problem(nx, ny, L, lmb=10.0, seed=42, f=None, data_term="square"):
    """BASELINE config 3: volumetric TV, gradient3d + sum_norm2(3) + sum_1d('square'); data_term='abs': volumetric TV-L1."""
    from . import block, function
    from .problem import variable, min_max_problem
    if f is None:
        f = rof_image(nx, ny, L, seed)
    u = variable(nx * ny * L)
    q = variable(3 * nx * ny * L)
    prob = min_max_problem([u], [q])
    prob.add_function(u, function.sum_1d(data_term, 1, f, lmb))
    prob.add_function(q, function.sum_norm2(3, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, block.gradient3d(nx, ny, L))
    return prob, u, q, f


def warp_matrix(N, seed_base=0):
    """A gather-type warp matrix W (n x 2n, n = N^2, exactly 4 non-zeros per row at DISPLACED columns): the linearised brightness-constancy
    term of TV-L1 optical flow, rho(u) = Ix(p + d) u1(p + d) + Iy(p + d) u2(p + d) - b, evaluated at the position p + d(p) a smooth
    warm-start flow d (|d| <= 5 pixels) points to, with LINEAR interpolation between the two neighbouring pixels along the flow's
    dominant axis per channel:  row p holds  Ix(q0) (1 - a), Ix(q1) a  at the columns of channel 1 and  Iy(r0) (1 - c), Iy(r1) c  at
    the columns of channel 2, q0 / q1 the two x-neighbours of the warped position (y rounded), r0 / r1 its two y-neighbours (x rounded).
    A general CSR block for block_sparse.cu:146-211 (the reference has no optical-flow example; example_tvl1.m:26-43 is the nearest
    problem).  -> scipy CSC matrix; Ix, Iy from the counter-hash generator (rof_image seeds 1, 2), column-major pixels (idx = y + x N)."""
    import scipy.sparse as sp
    n = N * N
    Ix = rof_image(N, N, 1, seed_base + 1) - 0.5
    Iy = rof_image(N, N, 1, seed_base + 2) - 0.5
    idx = np.arange(n)
    py, px = idx % N, idx // N
    dx = 4.5 * np.sin(2 * np.pi * 1.5 * px / N) * np.cos(2 * np.pi * py / N)
    dy = 4.5 * np.cos(2 * np.pi * px / N) * np.sin(2 * np.pi * 2.0 * py / N)
    qx = np.clip(px + dx, 0, N - 1); qy = np.clip(py + dy, 0, N - 1)
    x0 = np.minimum(np.floor(qx).astype(np.int64), N - 2); ax = qx - x0
    y0 = np.minimum(np.floor(qy).astype(np.int64), N - 2); ay = qy - y0
    xr = np.rint(qx).astype(np.int64); yr = np.rint(qy).astype(np.int64)
    c1a, c1b = yr + x0 * N, yr + (x0 + 1) * N                  # channel 1: the two x-neighbours at the rounded row
    c2a, c2b = y0 + xr * N, (y0 + 1) + xr * N                  # channel 2: the two y-neighbours at the rounded column
    rows = np.concatenate([idx, idx, idx, idx])
    cols = np.concatenate([c1a, c1b, n + c2a, n + c2b])
    # (weights kept away from exact zeros: every row has exactly 4 stored entries)
    vals = np.concatenate([Ix[c1a] * np.maximum(1 - ax, 1e-3), Ix[c1b] * np.maximum(ax, 1e-3), Iy[c2a] * np.maximum(1 - ay, 1e-3), Iy[c2b] * np.maximum(ay, 1e-3)])
    vals = np.where(vals == 0, 1e-3, vals)
    return sp.csc_matrix((vals, (rows, cols)), shape=(n, 2 * n))


def tvl1_flow_problem(N, warp=False):
    """BASELINE config 4: primal u in R^(2n); v = W u, g = gradient2d(N, N, 2) u; f(v) = sum_1d('abs', 1, b, 5), f(g) = sum_norm2(4, false,
    'abs'); min_problem.  warp=False: W = [diag(Ix) diag(Iy)] (SURVEY 8d C4, 2 non-zeros per row on the pixel's own columns);
    warp=True: the gather-type warp matrix above (4 non-zeros per row at displaced columns)"""
    import scipy.sparse as sp
    from . import block, function
    from .problem import variable, min_problem
    n = N * N
    bvec = rof_image(N, N, 1, 3) - 0.5
    if warp:
        W = warp_matrix(N)
    else:
        W = sp.hstack([sp.diags(rof_image(N, N, 1, 1) - 0.5), sp.diags(rof_image(N, N, 1, 2) - 0.5)]).tocsc()
    u = variable(2 * n)
    v, g = variable(n), variable(4 * n)
    prob = min_problem([u], [v, g])
    prob.add_function(v, function.sum_1d("abs", 1, bvec, 5.0))
    prob.add_function(g, function.sum_norm2(4, False, "abs"))
    prob.add_constraint(u, v, block.sparse(W))
    prob.add_constraint(u, g, block.gradient2d(N, N, 2))
    return prob
