"""Function (prox) builders -- Python mirror of matlab/+prost/+function/*.m.

Each builder returns ``func(idx, count) -> [name, idx, size, diagsteps, data]`` like the MATLAB
closures (sum_1d.m:79-80, sum_norm2.m:85-86, conjugate.m:7-15, sum_ind_epi_quad.m:17-20,
zero.m:3).  Models h(x) = c f(ax - b) + dx + 0.5 e x^2.
"""
import numpy as np

FUNCTIONS_1D = ("zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01",
                "max_pos0", "l0", "huber", "lq", "lq_plus_eps", "trunclin", "truncquad")


def _coeff(v):
    return np.atleast_1d(np.asarray(v, dtype=np.float64)).ravel()


def sum_1d(fun, a=1, b=0, c=1, d=0, e=0, alpha=0, beta=0):
    coeffs = [_coeff(v) for v in (a, b, c, d, e, alpha, beta)]
    return lambda idx, count: ["elem_operation:1d:" + fun, idx, count, True,
                               [count, 1, False, coeffs]]


def sum_norm2(dim, interleaved, fun, a=1, b=0, c=1, d=0, e=0, alpha=0, beta=0):
    coeffs = [_coeff(v) for v in (a, b, c, d, e, alpha, beta)]
    return lambda idx, count: ["elem_operation:norm2:" + fun, idx, count, False,
                               [count // dim, dim, bool(interleaved), coeffs]]


def conjugate(fun):
    def make(idx, count):
        child = fun(idx, count)
        return ["moreau", child[1], child[2], child[3], [child]]
    return make


def sum_ind_epi_quad(dim, interleaved, a, b, c):
    coeffs = [_coeff(a), _coeff(b), _coeff(c)]
    return lambda idx, count: ["ind_epi_quad", idx, count, False,
                               [count // dim, dim, bool(interleaved), coeffs]]


def zero():
    return lambda idx, count: ["zero", idx, count, True, []]


def transform(fun, a=1, b=0, c=1, d=0, e=0):
    """transform.m:1-50: c f(ax - b) + dx + (e/2) x^2 around ANY function `fun`"""
    coeffs = [_coeff(v) for v in (a, b, c, d, e)]

    def make(idx, count):
        child = fun(idx, count)
        return ["transform", child[1], child[2], child[3], coeffs + [child]]
    return make


def permute(fun, perm):
    """permute.m:1-17: composition with a permutation (local, 0-based indices)"""
    perm = np.asarray(perm, dtype=np.int64).ravel()

    def make(idx, count):
        child = fun(idx, count)
        return ["permute", child[1], child[2], child[3], [child, perm]]
    return make


def sum_ind_halfspace(dim, interleaved, a, b):
    """sum_ind_halfspace.m:1-19: projection onto a^T x <= b per group (a: dim or dim*count, b: 1 or count)"""
    coeffs = [_coeff(a), _coeff(b)]
    return lambda idx, count: ["ind_halfspace", idx, count, False,
                               [count // dim, dim, bool(interleaved), coeffs]]


def sum_ind_soc(dim, interleaved, alpha):
    """sum_ind_soc.m:1-20: projection onto alpha ||x|| <= y, variables ordered (x_1, .., x_{d-1}, y) planar"""
    return lambda idx, count: ["ind_soc", idx, count, False,
                               [count // dim, dim, bool(interleaved), float(alpha)]]


def sum_ind_sum(dim, interleaved):
    """sum_ind_sum.m:1-9: sum-to-one constraint per group"""
    return lambda idx, count: ["elem_operation:ind_sum", idx, count, False,
                               [count // dim, dim, bool(interleaved)]]


def sum_ind_simplex(dim, interleaved):
    """sum_ind_simplex.m:1-9: indicator of the unit simplex per group"""
    return lambda idx, count: ["elem_operation:ind_simplex", idx, count, False,
                               [count // dim, dim, bool(interleaved)]]


def sum_ind_sum2(dim, inds, s1, dim2=None, inds2=None, s2=None):
    """sum_ind_sum2.m:1-13: sum constraints over index arrays (one or two families)"""
    inds = np.asarray(inds, dtype=np.int64).ravel()
    if dim2 is None:
        return lambda idx, count: ["ind_sum", idx, count, True, [int(dim), inds, float(s1)]]
    inds2 = np.asarray(inds2, dtype=np.int64).ravel()
    return lambda idx, count: ["ind_sum", idx, count, True, [int(dim), inds, float(s1), int(dim2), inds2, float(s2)]]
