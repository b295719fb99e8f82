"""prost_amd -- MI355X-native primal-dual solver with the prost front-end surface.

Python mirror of the MATLAB package matlab/+prost (the reference's only front-end): the same
builders (variable, min_max_problem, block.*, function.*, backend.*, options) produce the same
nested problem descriptions, which are handed to the native solver (libprost.so, host C++ over
the HIP kernel C ABI libprost_hip.so) through the command table of include/prost_c.h
(`solve_problem`, `eval_linop`, `eval_prox`, ... -- matlab/+prost/private/prost.cpp:305-313).
"""
from . import backend, block, function  # noqa: F401
from .problem import (get_all_variables, min_max_problem, min_problem, options, problem,  # noqa: F401
                      sub_variable, variable)

__all__ = ["backend", "block", "function", "variable", "sub_variable", "problem", "min_max_problem",
           "min_problem", "options", "get_all_variables", "solve", "eval_linop", "eval_prox", "init", "release", "set_gpu",
           "list_gpus", "set_precision", "get_precision"]


def __getattr__(name):
    # the native command surface is imported lazily so that the pure description builders work
    # without the shared libraries (e.g. when only building problem descriptions)
    if name in ("solve", "eval_linop", "eval_prox", "init", "release", "set_gpu", "list_gpus",
                "set_precision", "get_precision", "problem_info", "Solver", "set_quirks", "comm_unique_id",
                "comm_init", "comm_init_host", "comm_destroy", "comm_info", "gloo_p2p", "load_plugin", "registered", "set_stop_callback", "set_output_callback", "ProstError"):
        from . import _capi
        return getattr(_capi, name)
    raise AttributeError(name)
