"""Backend option builders -- mirror of matlab/+prost/+backend/{pdhg,admm}.m."""


def pdhg(**kw):
    p = dict(tau0=1, sigma0=1, residual_iter=1, scale_steps_operator=True, alg2_gamma=0,
             arg_alpha0=0.5, arg_nu=0.95, arg_delta=1.5, arb_delta=1.05, arb_tau=0.8,
             stepsize="boyd")                                   # pdhg.m:4-14
    # MI355X addition: arithmetic="fmad" lets the fused iteration kernels contract multiply-adds and use fp32 reciprocal instructions
    # (iterates within a stated tolerance of the default "exact", which matches the CPU oracle bit for bit)
    arithmetic = kw.pop("arithmetic", None)
    _update(p, kw)
    if arithmetic is not None:
        if arithmetic not in ("exact", "fmad"):
            raise ValueError("arithmetic must be 'exact' or 'fmad'.")
        p["arithmetic"] = arithmetic
    return ["pdhg", p]


def admm(**kw):
    p = dict(rho0=1, residual_iter=1, arb_delta=1.05, arb_tau=0.8, arb_gamma=1.01, alpha=1.7,
             cg_max_iter=10, cg_tol_pow=1.3, cg_tol_min=1e-5, cg_tol_max=1e-8)   # admm.m:4-13
    _update(p, kw)
    return ["admm", p]


def _update(p, kw):
    for k, v in kw.items():
        if k not in p:
            raise ValueError("'%s' is not a recognized parameter." % k)   # inputParser behaviour
        p[k] = v
