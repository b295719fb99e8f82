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
