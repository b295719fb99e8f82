"""matlab/examples/example_multilabel_callback.m: the intermediate-solution callback of the two multilabel examples (example_multilabel_fast.m:60-63,
example_multilabel_tight.m:103-106).  The MATLAB function reads the labelling out of the iterate with prost.get_all_variables, shows it beside
the input image (imshow) and never asks the solver to stop; here the picture is handed to `show` (default: nothing) instead of a window."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost


def multilabel_callback(it, x, y, ny, nx, L, im, show=None):
    result = {"x": x, "y": y}                                                 # :4-5
    u = prost.variable(nx * ny * L)                                           # :6
    prost.get_all_variables(result, [u], [], [], [])                          # :7
    lab = np.asarray(u.val).reshape(L, nx, ny)                                # :9   reshape(u.val, [ny nx L])
    if show is not None:
        show(it, im, lab)                                                     # :10  imshow([im, u])
    return False                                                              # :12  is_converged = false
