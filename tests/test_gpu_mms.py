"""Manufactured solutions (the reference's own convergence study, examples/mms.py with examples/mmsldc2d): the discrete
solution of the GPU solve loop against the EXACT solution of the continuous Navier-Stokes problem -- the one check in this
suite that involves neither the oracle nor the product's own generator on both sides.  Expected orders: [P2]^2-P0 is
limited by the piecewise constant pressure (velocity L2 error O(h^2), gradient O(h), pressure O(h)); Scott-Vogelius
[P2]^2-P1dg on Alfeld splits is pressure-robust and of full order (O(h^3), O(h^2), O(h^2)), exactly divergence-free.
-m gpu"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.parametrize("disc,rates", [("pkp0", {"velocity": 1.8, "velocitygrad": 0.85, "pressure": 0.85}),
                                        ("sv", {"velocity": 2.7, "velocitygrad": 1.8, "pressure": 1.7})])
def test_convergence_orders_2d(disc, rates):
    from mms import study
    from alfi_amd.mms import convergence_orders
    hs, out = study(2, 4, [1, 2, 3], 2, disc, [1.0, 100.0], verbose=False)
    for re in (1.0, 100.0):
        for name, want in rates.items():
            orders = convergence_orders(out[re][name])
            assert orders[-1] > want, (disc, re, name, out[re][name], orders)
        if disc == "sv":       # div [P2]^2 is contained in the pressure space: exactly divergence-free
            assert max(out[re]["divergence"]) < 1e-7, out[re]["divergence"]


@pytest.mark.parametrize("k,rates", [(1, {"velocity": 1.6, "velocitygrad": 0.9, "pressure": 0.9}),
                                     (2, {"velocity": 1.7, "velocitygrad": 0.9, "pressure": 0.85})])
def test_convergence_orders_3d_on_the_baseline_elements(k, rates):
    """examples/mmsldc3d: [P1+FB]^3-P0 (config 3's element) and [P2+FB]^3-P0 (config 4's) on [0, 2]^3, N = 4 -> 8, Re = 1:
    second order for the velocity, first order for its gradient and for the pressure."""
    from mms import study
    from alfi_amd.mms import convergence_orders
    hs, out = study(3, 2, [1, 2], k, "pkp0", [1.0], verbose=False)
    for name, want in rates.items():
        orders = convergence_orders(out[1.0][name])
        assert orders[-1] > want, (k, name, out[1.0][name], orders)


def test_convergence_orders_3d_scott_vogelius_p3():
    """Config 5's discretisation -- [P3]^3 - P2dg on Alfeld splits, macro-star patches with condensed factors, macro-cell
    transfer blocks -- against the exact solution: orders towards 4 / 3 / 3 and a divergence at rounding level."""
    from mms import study
    from alfi_amd.mms import convergence_orders
    hs, out = study(3, 1, [1, 2], 3, "sv", [1.0], verbose=False)
    want = {"velocity": 3.4, "velocitygrad": 2.4, "pressure": 2.3}
    for name, w in want.items():
        orders = convergence_orders(out[1.0][name])
        assert orders[-1] > w, (name, out[1.0][name], orders)
    assert max(out[1.0]["divergence"]) < 1e-9
