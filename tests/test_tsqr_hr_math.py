"""The algorithm of csrc/fit_tsqr_kernel.hip (round 6), pinned on the CPU: TSQR over row chunks + Householder reconstruction reproduces LAPACK's own
reflectors / compact-WY T / R (dgeqrf, dlarft -- the convention Julia's qr uses, reference src/woodbury.jl:203), and the kernel's row-local formulas give
Vh and the mean (src/mvnormal.jl:14-21) without a sweep over the block.  NumPy statement: pathfinder.jl_amd/tools/tsqr_hr_check.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathfinder.jl_amd", "tools"))


@pytest.mark.parametrize("d,m,CH", [(1000, 12, 256), (3000, 8, 1024), (2500, 20, 512), (6000, 20, 1536), (2100, 32, 1024)])
def test_tsqr_plus_householder_reconstruction_equals_lapack(d, m, CH):
    import tsqr_hr_check as th
    r = th.check_reconstruction(d, m, CH)                      # includes a column of condition ~1e3
    assert r["T_lower"] == 0.0                                  # T comes out upper triangular by construction
    assert r["V"] <= 1e-11 and r["T"] <= 1e-10 and r["R"] <= 1e-12 and r["head"] <= 1e-10 and r["Q"] <= 1e-10, r
    w = th.check_reconstruction(d, m, CH, seed=3, ill=False)    # well conditioned: machine precision
    assert max(w["V"], w["T"], w["R"], w["head"], w["Q"]) <= 1e-13, w


@pytest.mark.parametrize("d,m,CH", [(1000, 12, 256), (2500, 20, 512), (5000, 20, 1536), (1800, 4, 4096)])
def test_row_local_formulas_of_the_kernel(d, m, CH):
    import tsqr_hr_check as th
    r = th.check_kernel_dataflow(d, m, CH)
    assert r["V"] <= 1e-13 and r["T"] <= 1e-13 and r["R"] <= 1e-13 and r["mu"] <= 1e-12, r
