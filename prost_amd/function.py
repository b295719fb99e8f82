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
