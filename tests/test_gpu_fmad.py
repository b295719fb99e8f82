"""The tolerance-class arithmetic (`backend.pdhg(..., arithmetic="fmad")`, prost_hip_fused_desc.arith = PROST_HIP_ARITH_FMAD).

The default ("exact") kernels round like the reference's expressions evaluated WITHOUT contraction and equal the CPU oracle bit for
bit.  The reference's own CUDA build contracts multiply-adds (nvcc's default, /root/reference/src/CMakeLists.txt:12-24), so its GPU
iterates are themselves only within rounding of that oracle; "fmad" is that class: fused multiply-adds where a product feeds a sum,
x / (1 + step) as a product with the fp32 reciprocal, pr v / ||v|| through v_rsq_f32 (elem_operation_1d.hpp:36-59,
elem_operation_norm2.hpp:40-88, backend_pdhg.cu:38-70).  Stated bounds, checked here against the exact
kernels / the oracle:

  * one iteration from identical inputs:  |x - x_exact| <= 2 ulp and |y - y_exact| <= 4 ulp AT THE VECTOR'S SCALE (ulp(max |v|)): the
    expressions cancel (v - b, differences of neighbours), so an error of one ulp of an operand is many ulps of a small result --
    a per-element ulp bound cannot hold for any contraction, nvcc's included; the per-element distribution is asserted in the bulk
    (99.9 % of the elements of x within 2 ulp of themselves);
  * iterates after k iterations:  max |v - v_oracle| / max |v_oracle| <= 1e-5 k  at k in {1, 10, 100} group iterations (measured:
    ~1e-7 .. 1e-6, the iteration is non-expansive);
  * a problem solved to 1e-4 stops within one residual period of where the exact solve stops;
  * residual norms within 1e-4 relative over the first ~100 iterations (later they approach the rounding noise of the iterates: 1e-2).

And exact identities of the class itself: a launch of K iterations equals every partition of it into shorter launches bit for bit
(K = 1 .. 6, any chunk length, any image size), which is what lets the host rebuild the iterate in front of a launch's last one.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import synthetic

pytestmark = pytest.mark.gpu

ZERO_TOL = dict(tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
TAUS = [0.3, 0.29, 0.28, 0.27, 0.26, 0.25, 0.24, 0.23, 0.22, 0.21, 0.2, 0.19]
SIGMAS = [1.0, 1.03, 1.06, 1.09, 1.12, 1.15, 1.18, 1.21, 1.24, 1.27, 1.3, 1.33]
THETAS = [0.9, 0.91, 0.92, 0.93, 0.94, 0.95, 0.96, 0.97, 0.98, 0.985, 0.99, 0.995]


@pytest.fixture(autouse=True)
def _gpu(hip):
    prost.set_gpu(0)
    prost.set_precision("single")
    yield
    prost.set_precision("double")


def _arr(v, i, k):
    return (C.c_double * k)(*v[i:i + k])


def _ulp_at(scale):
    return float(np.spacing(np.float32(scale)))


class _Rof:
    """device state of one ROF / TV-L1 problem at the kernel C ABI"""

    def __init__(self, hip, nx, ny, gfn="square", b_per_pixel=True, seed=0, lam=10.0, radius=1.0):
        self.hip, self.nx, self.ny = hip, nx, ny
        n, m = nx * ny, 2 * nx * ny
        rng = np.random.default_rng(seed)
        fh = synthetic.rof_image(nx, ny, seed=42).astype(np.float32)
        self.xh = (fh + 0.05 * (rng.random(n).astype(np.float32) - 0.5)).astype(np.float32)
        self.yh = ((rng.random(m) - 0.5) * 1.2).astype(np.float32)
        self.f = hip.DeviceArray.from_host(fh)
        self.x0 = hip.DeviceArray.from_host(self.xh); self.y0 = hip.DeviceArray.from_host(self.yh)
        self.x = [hip.DeviceArray.zeros(n, np.float32) for _ in range(2)]
        self.y = [hip.DeviceArray.zeros(m, np.float32) for _ in range(2)]
        self.gfn, self.bpp, self.lam, self.radius = gfn, b_per_pixel, lam, radius
        self.r4 = hip.DeviceArray.zeros(4, np.float64)
        self.ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)

    def desc(self, arith):
        hip = self.hip
        d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = self.nx, self.ny, 1
        d.g_fn = hip.FN_ID[self.gfn]; d.f_fn = hip.FN_ID["ind_leq0"]
        gv = [1, 0.4, self.lam, 0, 0, 0, 0]; fv = [1, self.radius, 1, 0, 0, 0, 0]
        for i in range(7):
            d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
        if self.bpp:
            d.g_coeff_ptr[1] = self.f.ptr.value
        d.T_val, d.S_val = 0.25, 0.5
        d.arith = arith
        return d

    def run(self, ks, arith=1, cols=0, res_last=False, mid=None):
        """from (x0, y0): launches of |k| iterations each (k < 0: the pair kernel); the iterate and the sums of the last launch"""
        hip = self.hip; L_ = hip.lib()
        n, m = self.nx * self.ny, 2 * self.nx * self.ny
        hip.check(L_.prost_hip_memcpy_d2d(self.x[0].ptr, self.x0.ptr, n * 4, None)); hip.check(L_.prost_hip_memcpy_d2d(self.y[0].ptr, self.y0.ptr, m * 4, None))
        d = self.desc(arith); it = 0
        I2 = hip.fn("fused_iteration2", np.float32)
        for i, k in enumerate(ks):
            a, b = i % 2, (i + 1) % 2
            last = res_last and i == len(ks) - 1
            if k == -2:
                hip.check(I2(C.byref(d), self.x[b].ptr, self.y[b].ptr, self.x[a].ptr, self.y[a].ptr, mid[0].ptr if mid else None, mid[1].ptr if mid else None,
                             _arr(TAUS, it, 2), _arr(SIGMAS, it, 2), _arr(THETAS, it, 2), 0, self.r4.ptr if last else None, self.ws.ptr if last else None, None))
                it += 2
            else:
                hip.check(L_.prost_hip_fused_iterationk_f32(C.byref(d), k, self.x[b].ptr, self.y[b].ptr, self.x[a].ptr, self.y[a].ptr, _arr(TAUS, it, k), _arr(SIGMAS, it, k),
                                                            _arr(THETAS, it, k), cols, self.r4.ptr if last else None, self.ws.ptr if last else None, None))
                it += k
        hip.sync()
        e = len(ks) % 2
        return self.x[e].to_host().copy(), self.y[e].to_host().copy(), self.r4.to_host().copy()


@pytest.mark.parametrize("nx,ny,gfn,bpp", [(4096, 4096, "square", True), (1000, 1024, "square", False), (333, 520, "abs", True), (64, 64, "abs", False), (9, 8, "square", True)])
def test_a_launch_of_k_iterations_equals_every_partition_bit_for_bit(hip, nx, ny, gfn, bpp):
    """K = 1 .. 6 iterations per launch (one and two halo lanes per side), automatic and odd chunk lengths, against six launches of the
    tolerance-class PAIR kernel (kernels_fused_iter2.hip, template parameter FMAD) -- two independently written kernels"""
    p = _Rof(hip, nx, ny, gfn, bpp)
    assert hip.lib().prost_hip_fused_iterationk_max(C.byref(p.desc(1)), 0) == 6
    assert hip.lib().prost_hip_fused_iterationk_max(C.byref(p.desc(0)), 0) == 4            # the exact class: K <= 4 (its own test below)
    assert hip.lib().prost_hip_fused_iterationk_max(C.byref(p.desc(1)), 1) == 0            # fp64: no tolerance-class instances
    ref = p.run([-2] * 6, res_last=True)
    assert np.isfinite(ref[0]).all() and np.isfinite(ref[1]).all()
    cases = [([2] * 6, 0), ([3] * 4, 0), ([4] * 3, 0), ([4, 3, 3, 2], 0), ([5, 5, 2], 0), ([6, 6], 0), ([1, 4, 1, 6], 0), ([6, 6], 5), ([5, 5, 2], 100), ([4] * 3, 7), ([3] * 4, 1000), ([2] * 6, 1)]
    for ks, cols in cases:
        if nx * ny > 2 ** 22 and cols in (1, 5, 7):
            continue                # (chunk lengths far below the automatic one on the large image: covered by the smaller ones)
        got = p.run(ks, cols=cols, res_last=ks[-1] >= 2)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (ks, cols, int((got[0] != ref[0]).sum()), int((got[1] != ref[1]).sum()))
        if ks[-1] >= 2:     # the four sums: same terms, the summation order depends on the launch geometry
            assert np.allclose(got[2], ref[2], rtol=1e-12), (ks, cols, got[2], ref[2])
    # a description the K-iteration kernel refuses is refused loudly
    d = p.desc(1); d.ny = ny + 1
    if hip.lib().prost_hip_fused_iterationk_max(C.byref(d), 0) == 0:
        assert hip.lib().prost_hip_fused_iterationk_f32(C.byref(d), 2, p.x[1].ptr, p.y[1].ptr, p.x[0].ptr, p.y[0].ptr, _arr(TAUS, 0, 2), _arr(SIGMAS, 0, 2), _arr(THETAS, 0, 2), 0,
                                                        None, None, None) != 0


@pytest.mark.parametrize("n", [4096, 1024])
def test_one_iteration_is_within_two_ulp_at_the_vectors_scale(hip, n):
    """exact and tolerance-class pair launch from the same inputs; the stored middle iterate is the result of ONE iteration"""
    p = _Rof(hip, n, n)
    mids = {}
    outs = {}
    for arith in (0, 1):
        mid = (hip.DeviceArray.zeros(n * n, np.float32), hip.DeviceArray.zeros(2 * n * n, np.float32))
        outs[arith] = p.run([-2], arith=arith, res_last=True, mid=mid)
        mids[arith] = (mid[0].to_host().copy(), mid[1].to_host().copy())
    assert hip.lib().prost_hip_fused_iteration2_arith(C.byref(p.desc(1)), 0) == 1 and hip.lib().prost_hip_fused_iteration2_arith(C.byref(p.desc(0)), 0) == 0
    for name, k, (e, t), bound in (("x", 1, (mids[0][0], mids[1][0]), 2.0), ("y", 1, (mids[0][1], mids[1][1]), 4.0),
                                   ("x", 2, (outs[0][0], outs[1][0]), 4.0), ("y", 2, (outs[0][1], outs[1][1]), 8.0)):
        scale = float(np.abs(e).max())
        err = float(np.abs(e.astype(np.float64) - t).max())
        assert err <= bound * _ulp_at(scale), (name, k, err / _ulp_at(scale))
        assert not np.array_equal(e, t)                         # (the two classes do differ: the test compares what it means to)
    # per element, in ulps of the element: the bulk of x (no cancellation at its magnitude) is within 2 ulp
    e, t = mids[0][0], mids[1][0]
    u = np.abs(e.astype(np.float64) - t) / np.spacing(np.maximum(np.abs(e), np.float32(1e-30)))
    assert float(np.quantile(u, 0.999)) <= 2.0, float(np.quantile(u, 0.999))
    # residual sums of the second iteration
    assert np.allclose(outs[0][2], outs[1][2], rtol=1e-4), (outs[0][2], outs[1][2])


def _solve_pair(n, k, backend_kw, L=None, arithmetic="fmad"):
    if L is None:
        prob, u, q, f = synthetic.rof_problem(n, n)
    else:
        prob, u, q, f = synthetic.tv3d_problem(n, n, L)
    b = prost.backend.pdhg(arithmetic=arithmetic, **backend_kw)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    info = s.iterate(k, time_kernels=True, sample_every=1)
    st = s.state()
    s.destroy()
    return prob, b, o, st, info


@pytest.mark.parametrize("n,ks", [(4096, (3, 12, 102)), (1024, (12, 102, 1002)), (256, (102,))])
def test_c2_iterates_track_the_oracle(n, ks):
    """4096^2 is the headline size and launch geometry (groups of 4, 4 and 2 + residual sums per residual period of 10)"""
    for k in ks:
        prob, b, o, st, info = _solve_pair(n, k, dict(stepsize="alg2", residual_iter=10, alg2_gamma=0.5))
        assert st["path"] == "pdhg:fused-grad2d+fmad" and st["arithmetic"] == "fmad" and st["iterations_per_launch_max"] == 4
        names = set(info["kernels"])
        if k >= 12:
            assert any(nm.startswith("fused_iter2d_xk_kernel") for nm in names), names
            assert "fused_iter2d_xk_kernel<4>" in names and any(nm.endswith("+residuals") and "xk" in nm for nm in names), names
        oracle.set_num_threads(16)
        bo = [b[0], {kk: v for kk, v in b[1].items() if kk != "arithmetic"}]
        os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float32)
        os_.initialize(); os_.iterate(k)
        ost = os_.state(); sc = os_.scalars()
        del os_
        kf = max(1, k - 2)                                  # iterations 0 and 1 run the exact single launches
        for v in "xyzw":
            scale = float(np.abs(ost[v]).max())
            rel = float(np.abs(st[v].astype(np.float64) - ost[v]).max()) / scale
            assert rel <= 1e-5 * kf, (n, k, v, rel)
            assert np.isfinite(st[v]).all()
        for v in ("tau", "sigma", "theta"):
            assert st[v] == sc[v], v                        # alg2: the step sizes do not depend on the data
        # residual norms: differences of successive iterates -- late in a solve they shrink towards the rounding noise of the
        # iterates themselves (1002 iterations at 1024^2: 0.077 over 10^6 pixels), where two roundings of the same iteration differ
        # in the third digit; up to ~100 iterations they agree to 1e-4
        for v in ("primal_res", "dual_res"):
            assert np.isclose(st[v], sc[v], rtol=1e-4 if k <= 102 else 1e-2), (v, st[v], sc[v])


@pytest.mark.parametrize("rule,ri", [("boyd", 1), ("boyd", 10), ("goldstein", 7), ("alg1", 3), ("alg2", 4)])
def test_every_step_rule_and_residual_period_tracks_the_oracle(rule, ri):
    """residual-driven rules run on the device record (prost_hip_fused_iterationk_rec); residual_iter = 1 leaves no room for a group"""
    n, k = 512, 60
    prob, b, o, st, info = _solve_pair(n, k, dict(stepsize=rule, residual_iter=ri))
    assert st["iteration"] == k
    bo = [b[0], {kk: v for kk, v in b[1].items() if kk != "arithmetic"}]
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float32)
    os_.initialize(); os_.iterate(k)
    ost = os_.state(); sc = os_.scalars()
    del os_
    if ri > 1:
        assert any(nm.startswith("fused_iter2d_xk_kernel") for nm in info["kernels"]), info["kernels"]
    # (a residual-driven rule that decides differently on a knife edge would show as a step-size mismatch first)
    for v in ("tau", "sigma"):
        assert np.isclose(st[v], sc[v], rtol=1e-5), (rule, ri, v, st[v], sc[v])
    for v in "xyzw":
        rel = float(np.abs(st[v].astype(np.float64) - ost[v]).max()) / float(np.abs(ost[v]).max())
        assert rel <= 1e-5 * k, (rule, ri, v, rel)


def test_rof_1024_solved_to_tolerance_stops_within_one_residual_period_of_the_exact_solve():
    n = 1024
    prob, u, q, f = synthetic.rof_problem(n, n)
    res = {}
    for ar in ("exact", "fmad"):
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5, arithmetic=ar)
        o = prost.options(max_iters=20000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-4, tol_rel_dual=1e-4, tol_abs_primal=1e-4, tol_abs_dual=1e-4)
        res[ar] = prost.solve(prob, b, o)
        assert res[ar]["result"] == "Converged.", res[ar]["result"]
    assert abs(res["exact"]["iters"] - res["fmad"]["iters"]) <= 10, (res["exact"]["iters"], res["fmad"]["iters"])
    x0, x1 = np.asarray(res["exact"]["x"]).reshape(-1), np.asarray(res["fmad"]["x"]).reshape(-1)
    assert float(np.abs(x0 - x1).max()) <= 1e-4                 # two runs that may differ by a residual period, both at the minimiser to 1e-4


def test_callbacks_and_read_outs_rebuild_the_previous_iterate_in_the_same_class():
    """z and w need the iterate in front of the last one, which a group keeps in registers: RebuildPrevious re-runs g - 1 iterations
    through the K-iteration kernel.  (An observed run partitions the iterations differently -- a residual iteration left alone runs
    the exact single launch -- so the two final iterates agree to the class tolerance, not bit for bit.)"""
    n, k = 512, 44
    prob, u, q, f = synthetic.rof_problem(n, n)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5, arithmetic="fmad")
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o); s.iterate(k); plain = s.state(); s.destroy()
    s = prost.Solver(prob, b, o)
    seen = []
    for i in range(0, k, 4):
        s.iterate(4)
        seen.append(s.state())                                  # (reads x, y, z, w: rebuilds the previous iterate every time)
    last = seen[-1]
    s.destroy()
    for v in "xy":
        assert float(np.abs(plain[v].astype(np.float64) - last[v]).max()) <= 1e-5 * k * float(np.abs(plain[v]).max()), v
    bo = [b[0], {kk: vv for kk, vv in b[1].items() if kk != "arithmetic"}]
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float32)
    os_.initialize(); os_.iterate(k)
    ost = os_.state()
    del os_
    for v in "xyzw":
        rel = float(np.abs(last[v].astype(np.float64) - ost[v]).max()) / float(np.abs(ost[v]).max())
        assert rel <= 1e-5 * k, (v, rel)


def test_shapes_without_a_tolerance_class_instance_run_exact():
    """fp64, ragged heights, position-dependent Tau, other function pairs: `arithmetic="fmad"` is a permission, not a promise --
    the solve is then bit-identical to the oracle"""
    prost.set_precision("double")
    n, k = 256, 24
    prob, b, o, st, info = _solve_pair(n, k, dict(stepsize="alg2", residual_iter=10, alg2_gamma=0.5))
    assert st["arithmetic"] == "exact" and st["path"] == "pdhg:fused-grad2d"
    bo = [b[0], {kk: v for kk, v in b[1].items() if kk != "arithmetic"}]
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float64)
    os_.initialize(); os_.iterate(k)
    ost = os_.state()
    del os_
    for v in "xyzw":
        assert np.array_equal(st[v], ost[v]), v
    with pytest.raises(ValueError):
        prost.backend.pdhg(arithmetic="fast")


@pytest.mark.parametrize("kind,dims", [("3d", (48, 40, 9)), ("3d", (128, 96, 20)), ("rgb", (96, 128, 3)), ("rgb", (200, 64, 2)), ("rgb", (64, 64, 4))])
def test_volumes_and_colour_images_track_the_oracle(kind, dims):
    """the tolerance-class instances of the 3-D and the multi-channel pair kernels (two iterations per launch as in the exact class)"""
    nx, ny, L = dims
    k = 42
    f = synthetic.rof_image(nx, ny, L, 42)
    prob = (synthetic.tv3d_problem(nx, ny, L, f=f) if kind == "3d" else synthetic.rof_problem(nx, ny, L, f=f))[0]
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5, arithmetic="fmad")
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    info = s.iterate(k, time_kernels=True, sample_every=1)
    st = s.state()
    s.destroy()
    assert st["arithmetic"] == "fmad" and st["path"] == ("pdhg:fused-grad3d+fmad" if kind == "3d" else "pdhg:fused-grad2d+fmad"), (st["arithmetic"], st["path"])
    assert any("_x2_kernel" in nm for nm in info["kernels"]), info["kernels"]
    bo = [b[0], {kk: v for kk, v in b[1].items() if kk != "arithmetic"}]
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float32)
    os_.initialize(); os_.iterate(k)
    ost = os_.state(); sc = os_.scalars()
    del os_
    differs = False
    for v in "xyzw":
        rel = float(np.abs(st[v].astype(np.float64) - ost[v]).max()) / float(np.abs(ost[v]).max())
        assert rel <= 1e-5 * k, (kind, dims, v, rel)
        differs = differs or not np.array_equal(st[v], ost[v])
    assert differs                                  # (the tolerance-class instance did run)
    # (residual norms: a per-element residual of ~2e-4 is formed from terms of size 0.1 .. 1, so two roundings of the same iterate
    # differ by ~5e-4 of it)
    for v in ("primal_res", "dual_res"):
        assert np.isclose(st[v], sc[v], rtol=2e-3), (v, st[v], sc[v])


def _crop(f, nx, ny, L, x0, x1, y0, y1, l0, l1):
    return np.ascontiguousarray(f.reshape(L, nx, ny)[l0:l1, x0:x1, y0:y1]).reshape(-1)


def _read_block(solver, which, comps, nx, ny, L, x0, x1, y0, y1, l0, l1):
    n = nx * ny * L
    offs = [c * n + l * nx * ny + x * ny + y0 for c in range(comps) for l in range(l0, l1) for x in range(x0, x1)]
    return solver.read(which, offs, y1 - y0).reshape(comps, l1 - l0, x1 - x0, y1 - y0)


def test_c3_2048x2048x64_sub_volumes_track_the_oracle():
    """BASELINE config 3 at its full size in the tolerance class: sub-volumes of the 2048 x 2048 x 64 run against oracle runs on the
    cropped volume (tests/test_gpu_fullsize.py: after k iterations every voxel further than k from a crop's artificial faces depends
    on the crop's data only)"""
    nx, ny, L, k = 2048, 2048, 64, 12
    f = synthetic.rof_image(nx, ny, L, 42)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    prob = synthetic.tv3d_problem(nx, ny, L, f=f)[0]
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5, arithmetic="fmad")
    s = prost.Solver(prob, b, o)
    del prob
    info = s.iterate(k, time_kernels=True)
    scal = s.state(vectors=False)
    assert scal["path"] == "pdhg:fused-grad3d+fmad" and scal["iteration"] == k
    assert any(nm.startswith("fused_iter3d_x2_kernel") for nm in info["kernels"]), info["kernels"]
    oracle.set_num_threads(16)
    margin = k + 1
    bo = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    for (x0, x1, y0, y1, l0, l1) in [(nx - 44, nx, ny - 72, ny, L - 36, L), (1000, 1044, 1990, 2048, 20, 56)]:
        cx, cy, cl = x1 - x0, y1 - y0, l1 - l0
        cprob = synthetic.tv3d_problem(cx, cy, cl, f=_crop(f, nx, ny, L, x0, x1, y0, y1, l0, l1))[0]
        cprob.finalize()
        osv = oracle.Solver(cprob.data, cprob.nrows, cprob.ncols, bo, o, np.float32)
        osv.initialize(); osv.iterate(k)
        ost = osv.state()
        lo = lambda a0: 0 if a0 == 0 else margin
        hi = lambda a1, full, c: c if a1 == full else c - margin
        sx, sy, sl = slice(lo(x0), hi(x1, nx, cx)), slice(lo(y0), hi(y1, ny, cy)), slice(lo(l0), hi(l1, L, cl))
        got_x = _read_block(s, "x", 1, nx, ny, L, x0, x1, y0, y1, l0, l1)[:, sl, sx, sy]
        got_y = _read_block(s, "y", 3, nx, ny, L, x0, x1, y0, y1, l0, l1)[:, sl, sx, sy]
        exp_x = ost["x"].reshape(1, cl, cx, cy)[:, sl, sx, sy]
        exp_y = ost["y"].reshape(3, cl, cx, cy)[:, sl, sx, sy]
        assert got_x.size > 4000
        for name, g, e in (("x", got_x, exp_x), ("y", got_y, exp_y)):
            rel = float(np.abs(g.astype(np.float64) - e).max()) / float(np.abs(e).max())
            assert rel <= 1e-5 * k, (name, (x0, y0, l0), rel)
            assert not np.array_equal(g, e)
    s.destroy()


@pytest.mark.parametrize("nx,ny,gfn,bpp", [(1024, 1024, "square", True), (333, 520, "abs", True), (40, 64, "square", False)])
def test_the_k_iteration_kernel_in_the_exact_class_equals_the_exact_pair_kernel_bit_for_bit(hip, nx, ny, gfn, bpp):
    """the pipeline of kernels_fused_iterk.hip (stage order, halo lanes, LDS ring, chunk edges) with the EXACT forms: equal to the exact pair
    kernel -- which equals the CPU oracle (tests/test_gpu_fullsize.py) -- for K = 1 .. 4 and every partition"""
    p = _Rof(hip, nx, ny, gfn, bpp)
    assert hip.lib().prost_hip_fused_iterationk_max(C.byref(p.desc(0)), 0) == 4
    ref = p.run([-2] * 6, arith=0, res_last=True)
    for ks, cols in (([2] * 6, 0), ([3] * 4, 0), ([4] * 3, 0), ([4, 3, 3, 2], 0), ([1, 4, 1, 4, 2], 0), ([4] * 3, 5), ([3] * 4, 1000)):
        got = p.run(ks, arith=0, cols=cols, res_last=ks[-1] >= 2)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (ks, cols)
        if ks[-1] >= 2:
            assert np.allclose(got[2], ref[2], rtol=1e-12), (ks, cols)
    d = p.desc(0)
    assert hip.lib().prost_hip_fused_iterationk_f32(C.byref(d), 5, p.x[1].ptr, p.y[1].ptr, p.x[0].ptr, p.y[0].ptr, _arr(TAUS, 0, 5), _arr(SIGMAS, 0, 5), _arr(THETAS, 0, 5), 0,
                                                    None, None, None) != 0          # the exact class runs K <= 4


def test_exact_groups_through_the_solver_equal_the_oracle_bit_for_bit():
    """Options::group_max = 3 in the exact class (pairs are its default): the group scheduling, the rebuilt previous iterate and the
    residual launches against oracle.Solver, every bit of x, y, z, w"""
    n, k = 1024, 47
    prob, u, q, f = synthetic.rof_problem(n, n)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    b[1]["group_max"] = 3
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    info = s.iterate(k, time_kernels=True, sample_every=1)
    st = s.state()
    s.destroy()
    assert st["arithmetic"] == "exact" and st["path"] == "pdhg:fused-grad2d" and st["iterations_per_launch_max"] == 3
    assert "fused_iter2d_xk_kernel<3>" in info["kernels"], info["kernels"]
    bo = [b[0], {kk: v for kk, v in b[1].items() if kk != "group_max"}]
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, np.float32)
    os_.initialize(); os_.iterate(k)
    ost = os_.state()
    del os_
    for v in "xyzw":
        assert np.array_equal(st[v], ost[v]), (v, int((st[v] != ost[v]).sum()))
