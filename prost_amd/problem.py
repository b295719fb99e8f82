"""Problem description classes -- Python mirror of matlab/+prost/{variable,sub_variable,problem,
min_max_problem,min_problem}.m.

These are pure data builders: they assign 0-based offsets to variables and collect nested
lists (the MATLAB cell arrays) under ``prob.data`` with the same field names the MEX factory
reads (factory.cpp:950-990): linop, prox_g, prox_f, prox_gstar, prox_fstar, scaling,
scaling_alpha | scaling_left/right.
"""
import numpy as np


class variable:
    """variable.m:1-16"""

    def __init__(self, dim):
        self.dim = int(dim)
        self.val = np.zeros(self.dim)
        self.sub_vars = []
        self.idx = None


class sub_variable:
    """sub_variable.m:1-19"""

    def __init__(self, parent, dim):
        self.dim = int(dim)
        self.parent = parent
        self.val = np.zeros(self.dim)
        self.idx = None
        parent.sub_vars.append(self)


def _add_prox(prox_to_add, prox_list):
    """private/add_prox.m:1-19 -- replace a prox with the same idx, else append."""
    idx = prox_to_add[1]
    for i, p in enumerate(prox_list):
        if p[1] == idx:
            prox_list[i] = prox_to_add
            return prox_list
    prox_list.append(prox_to_add)
    return prox_list


class problem:
    """problem.m:1-36"""

    def __init__(self):
        self.data = {}
        self.set_scaling_alpha(1)
        self.data["prox_f"] = []
        self.data["prox_fstar"] = []
        self.data["prox_g"] = []
        self.data["prox_gstar"] = []
        self.data["linop"] = []
        self.nrows = 0
        self.ncols = 0

    def set_scaling_identity(self):
        self.data["scaling"] = "identity"
        return self

    def set_scaling_alpha(self, alpha):
        self.data["scaling"] = "alpha"
        self.data["scaling_alpha"] = alpha
        return self

    def set_scaling_custom(self, left, right):
        self.data["scaling"] = "custom"
        self.data["scaling_left"] = np.asarray(left, dtype=np.float64).ravel()
        self.data["scaling_right"] = np.asarray(right, dtype=np.float64).ravel()
        return self

    # -- shared helpers -------------------------------------------------------------------
    @staticmethod
    def _assign_indices(vars_):
        idx = 0
        for v in vars_:
            v.idx = idx
            sub_idx = 0
            for sv in v.sub_vars:
                sv.idx = idx + sub_idx
                sub_idx += sv.dim
            if sub_idx != v.dim and len(v.sub_vars) > 0:
                raise ValueError("Size of subvariables does not match size of parent variable.")
            idx += v.dim
        return idx

    @staticmethod
    def _find(vars_, var):
        """returns (idx, dim) of var or one of the sub variables, else None"""
        for v in vars_:
            for sv in v.sub_vars:
                if sv is var:
                    return sv.idx, sv.dim
            if v is var:
                return v.idx, v.dim
        return None

    def _add_block(self, pv_vars, dv_vars, pv, dv, block, what):
        c = self._find(pv_vars, pv)
        r = self._find(dv_vars, dv)
        if r is None or c is None:
            raise ValueError("Variable pair not registered in problem.")
        row, dual_dim = r
        col, primal_dim = c
        block_size_pair = block(row, col, dual_dim, primal_dim)
        for i, lin in enumerate(self.data["linop"]):
            if lin[1] == row and lin[2] == col:
                self.data["linop"][i] = block_size_pair[0]      # constraint is replaced
                break
        else:
            self.data["linop"].append(block_size_pair[0])
        sz = block_size_pair[1]
        if sz[0] != dual_dim or sz[1] != primal_dim:
            raise ValueError("Size of block does not fit size of primal/%s variable." % what)
        return self

    @staticmethod
    def _fill(vars_, vec):
        for v in vars_:
            v.val = np.array(vec[v.idx:v.idx + v.dim])
            for sv in v.sub_vars:
                a = sv.idx - v.idx
                sv.val = v.val[a:a + sv.dim]


class min_max_problem(problem):
    """min_max_problem.m:1-229: min_x max_y g(x) + <Kx,y> - f*(y)"""

    def __init__(self, primals, duals):
        super().__init__()
        self.primal_vars = list(primals)
        self.dual_vars = list(duals)
        self.num_primal_vars = len(self.primal_vars)
        self.num_dual_vars = len(self.dual_vars)
        self.ncols = self._assign_indices(self.primal_vars)
        self.nrows = self._assign_indices(self.dual_vars)

    def add_function(self, var, func):
        hit = self._find(self.primal_vars, var)
        if hit is not None:
            self.data["prox_g"] = _add_prox(func(hit[0], hit[1]), self.data["prox_g"])
            return self
        hit = self._find(self.dual_vars, var)
        if hit is not None:
            self.data["prox_fstar"] = _add_prox(func(hit[0], hit[1]), self.data["prox_fstar"])
            return self
        raise ValueError("Variable not registered in problem!")

    def add_dual_pair(self, pv, dv, block):
        return self._add_block(self.primal_vars, self.dual_vars, pv, dv, block, "dual")

    def fill_variables(self, result):
        self._fill(self.primal_vars, result["x"])
        self._fill(self.dual_vars, result["y"])
        return self

    def finalize(self):
        from .function import zero
        zero_fn = zero()
        if not self.data["prox_g"]:
            self.data["prox_g"].append(zero_fn(0, self.ncols))
        if not self.data["prox_fstar"]:
            self.data["prox_fstar"].append(zero_fn(0, self.nrows))
        return self


class min_problem(problem):
    """min_problem.m:1-228: min_{x,z} g(x) + f(z) s.t. z = Kx"""

    def __init__(self, primals, constraineds):
        super().__init__()
        self.primal_vars = list(primals)
        self.constrained_vars = list(constraineds)
        self.num_primal_vars = len(self.primal_vars)
        self.num_constrained_vars = len(self.constrained_vars)
        self.ncols = self._assign_indices(self.primal_vars)
        self.nrows = self._assign_indices(self.constrained_vars)

    def add_function(self, var, func):
        hit = self._find(self.primal_vars, var)
        if hit is not None:
            self.data["prox_g"] = _add_prox(func(hit[0], hit[1]), self.data["prox_g"])
            return self
        hit = self._find(self.constrained_vars, var)
        if hit is not None:
            self.data["prox_f"] = _add_prox(func(hit[0], hit[1]), self.data["prox_f"])
            return self
        raise ValueError("Variable not registered in problem!")

    def add_constraint(self, pv, cv, block):
        return self._add_block(self.primal_vars, self.constrained_vars, pv, cv, block, "constrained")

    def fill_variables(self, result):
        self._fill(self.primal_vars, result["x"])
        self._fill(self.constrained_vars, result["z"])
        return self

    def finalize(self):
        from .function import zero
        zero_fn = zero()
        if not self.data["prox_g"]:
            self.data["prox_g"].append(zero_fn(0, self.ncols))
        if not self.data["prox_f"]:
            self.data["prox_f"].append(zero_fn(0, self.nrows))
        return self


def get_all_variables(result, p_vars, pc_vars, d_vars, dc_vars):
    """matlab/+prost/get_all_variables.m:19-49: slices result x / z / y / w into the given primal, constrained-primal, dual and
    constrained-dual variables, in order (sub-variables are not set, as in the reference)"""
    import numpy as np
    for key, vs in (("x", p_vars), ("z", pc_vars), ("y", d_vars), ("w", dc_vars)):
        idx = 0
        for v in vs:
            v.val = np.asarray(result[key]).reshape(-1)[idx:idx + v.dim]
            idx += v.dim


def options(**kw):
    """options.m:4-14"""
    p = dict(tol_rel_primal=1e-4, tol_rel_dual=1e-4, tol_abs_primal=1e-4, tol_abs_dual=1e-4,
             max_iters=1000, num_cback_calls=10, verbose=True, interm_cb=None,
             x0=None, y0=None, solve_dual=False)
    for k, v in kw.items():
        if k not in p:
            raise ValueError("'%s' is not a recognized parameter." % k)
        p[k] = v
    return p
