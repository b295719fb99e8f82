"""The reference's example scripts, mirrored in examples/, run end to end on the MI355X (reduced sizes)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import prost_amd as prost

pytestmark = pytest.mark.gpu


def test_example_rof_primaldual_converges_by_its_own_gap_callback():
    import rof_rgb_gap_callback as ex
    prost.set_gpu(0)
    prost.set_precision("double")          # the MEX default (config.hpp:7)
    result, gaps, img = ex.main(nx=70, ny=48, nc=3, max_iters=10000, verbose=False)
    assert result["result"] == "Stopped by user." or result["result"] == "Converged."
    assert gaps and gaps[-1] < 1e-5 and gaps[-1] <= gaps[0]
    assert img.shape == (3, 70, 48) and np.isfinite(img).all()


def test_example_tvl1_removes_salt_and_pepper_noise():
    import tvl1_salt_and_pepper as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    result, err_noisy, err_denoised = ex.main(nx=96, ny=64, nc=1, max_iters=20000, verbose=False)
    assert result["result"] in ("Converged.", "Reached maximum iterations.")
    assert err_denoised < 0.5 * err_noisy
