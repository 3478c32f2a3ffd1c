"""Soft pin against the only numbers the reference holds for this path: the recorded outputs of
its two notebooks.  The notebooks' data are UNSEEDED draws, bayesianquilts' loss scaling, surrogate
parameterisation and waic() are out of tree ([UNVERIFIED-3P]); so this is a statistical check
that the build-defined ELBO scaling and waic() land where the reference's
did -- it does not pin the oracle ("parity unpinned" stands).

  notebooks/factorizing_random_noise.ipynb:51-62,122,447   loss 44.13 -> 40.39, lppd -37090.95
  notebooks/factorize_linear_structure.ipynb:53-66,447,468  loss 54.51 -> 46.97, lppd -41236.9

Measured here (round 3, seeds below): noise 39.72 (-1.7 %), lppd -38 777 (4.5 % lower);
linear 46.10 (-1.9 %), lppd -44 726 (8.5 % lower).  The final loss is held to +-5 %.  The lppd is
held to +-10 %: it is log mean_s p(x | theta_s) over surrogate draws, and the reference's draws are
far more dispersed than this build's (its pwaic is 135 483 / 180 388 against 10 / 2.6 here: the
variance of the per-cell log-likelihood over draws), which raises a log-mean-exp; the surrogate's
parameterisation is exactly the part that is build-defined."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

REFERENCE = {   # notebook cell outputs (file:line above)
    "noise": {"first_loss": 44.13152280855831, "final_loss": 40.3874, "lppd": -37090.95152008469},
    "linear": {"first_loss": 54.505081021975975, "final_loss": 46.9736, "lppd": -41236.92593508755},
}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("which", ["noise", "linear"])
def test_notebook_recorded_outputs_are_reproduced_statistically(which):
    import notebook_pins
    got = notebook_pins.run(which, epochs=200, seed=0)
    ref = REFERENCE[which]
    assert got["final_loss"] < got["first_loss"]
    assert abs(got["final_loss"] - ref["final_loss"]) <= 0.05 * ref["final_loss"], got
    assert abs(got["lppd"] - ref["lppd"]) <= 0.10 * abs(ref["lppd"]), got
    # both quantities are per-batch-row-scale numbers: the lppd of 1000 x 30 cells
    assert -60_000 < got["lppd"] < -30_000
