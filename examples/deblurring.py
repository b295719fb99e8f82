"""matlab/examples/example_deblurring.m on the MI355X build, line for line: TV deblurring in the constrained (primal) form --
min_u lmb/2 |B u - f_blurred|^2 + |grad u|_{2,1} with v = B u and g = grad u as constrained variables (prost.min_problem, :29-37), both
operators handed over as SPARSE MATRICES (B = kron(speye(nc), convmtx2(kernel, ny, nx)), :14-17; spmat_gradient2d, :10), the default
backend options boyd / residual_iter 1 (:40-41).  The script's commented-out lines :12-24 (motion kernel, full 2-D convolution matrix,
blurred + noisy data) are restated below; synthetic image instead of images/flowers.png.  usage: python examples/deblurring.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import prost_amd as prost
from multilabel_fast import spmat_gradient2d
from prost_amd import synthetic


def motion_kernel(length=15, angle_deg=45.0):
    """fspecial('motion', 15, 45) in spirit: a normalised anti-aliased line segment through the centre of an odd square window"""
    half = (length - 1) / 2.0
    c, s_ = np.cos(np.deg2rad(angle_deg)), np.sin(np.deg2rad(angle_deg))
    r = int(np.ceil(half * max(abs(c), abs(s_))))
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1].astype(np.float64)
    dist = np.abs(xx * s_ + yy * c)                      # distance from the line (image rows grow downwards)
    along = np.abs(xx * c - yy * s_)
    k = np.clip(1.0 - dist, 0.0, None) * (along <= half + 0.5)
    return k / k.sum()


def convmtx2(kernel, ny, nx):
    """convmtx2(kernel, ny, nx): the matrix of the FULL 2-D convolution of a column-major ny x nx image, (ny + ky - 1)(nx + kx - 1) rows"""
    ky, kx = kernel.shape
    ny2, nx2 = ny + ky - 1, nx + kx - 1
    rows, cols, vals = [], [], []
    yy, xx = np.mgrid[0:ny, 0:nx]
    src = (yy + xx * ny).reshape(-1)
    for j in range(kx):
        for i in range(ky):
            if kernel[i, j] == 0:
                continue
            rows.append(((yy + i) + (xx + j) * ny2).reshape(-1)); cols.append(src); vals.append(np.full(src.size, kernel[i, j]))
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ny2 * nx2, ny * nx))


def describe(nx, ny, nc, lmb=100.0, klen=15, tol=1e-4, max_iters=25000, num_cback_calls=250, seed=9):
    f = synthetic.rof_image(nx, ny, nc, seed).astype(np.float64)                              # :2-6
    grad = spmat_gradient2d(nx, ny, nc)                                                       # :10
    kernel = motion_kernel(klen, 45.0)                                                        # :13
    B = sp.kron(sp.eye(nc), convmtx2(kernel, ny, nx)).tocsc()                                 # :15-16
    ky, kx = kernel.shape                                                                     # :18-19
    nx2, ny2 = nx + kx - 1, ny + ky - 1                                                       # :21-22
    rng = np.random.default_rng(7)
    f_blurred = B @ f + 0.05 * rng.standard_normal(ny2 * nx2 * nc)                            # :24-25
    u = prost.variable(nx * ny * nc)                                                          # :29
    v = prost.variable(nx2 * ny2 * nc)                                                        # :30
    g = prost.variable(2 * nx * ny * nc)                                                      # :31
    prob = prost.min_problem([u], [v, g])                                                     # :33
    prob.add_function(v, prost.function.sum_1d("square", 1, f_blurred, lmb, 0, 0))            # :34
    prob.add_function(g, prost.function.sum_norm2(2 * nc, False, "abs", 1, 0, 1, 0, 0))       # :35
    prob.add_constraint(u, v, prost.block.sparse(B))                                          # :36
    prob.add_constraint(u, g, prost.block.sparse(grad))                                       # :37
    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=1)                            # :40-41
    opts = prost.options(max_iters=max_iters, num_cback_calls=num_cback_calls, verbose=False, tol_rel_primal=tol, tol_abs_primal=tol,
                         tol_rel_dual=tol, tol_abs_dual=tol)                                  # :44-50
    return prob, backend, opts, u, f, f_blurred, (nx2, ny2)


def main(nx=256, ny=192, nc=3, max_iters=25000, verbose=True, backend_opts=None, klen=15):
    prob, backend, opts, u, f, f_blurred, _ = describe(nx, ny, nc, max_iters=max_iters, klen=klen)
    if backend_opts:
        backend[1].update(backend_opts)
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                                 # :53
    elapsed = time.perf_counter() - t0
    img = np.asarray(u.val)
    if verbose:
        print("%s after %d iterations, %.3f s on %s; |u - f| mean %.4f" % (result["result"], result["iters"], elapsed, result["path"], np.abs(img - f).mean()))
    return result, img, f


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
